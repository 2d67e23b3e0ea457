# A/B of the stage implementations above D = 64 (large_d.hip) on one box: the fused sweep per size and version.
run() { echo "FUSED=$1 WIDE=$2 : $3"; VGPA_STAGE_FUSED=$1 VGPA_STAGE_WIDE=$2 python tools/bench_large_d_sweep.py $3; }
for cfg in "72 401 1" "96 401 1" "128 401 1" "200 401 1" "256 401 1" "384 201 1" "512 201 1" "128 401 8"; do
  run 0 0 "$cfg"; run 4096 0 "$cfg"; run 0 4096 "$cfg"
done
for cfg in "640 101 1" "768 101 1" "1000 81 1" "1024 81 1" "1536 41 1" "2048 21 1"; do
  run 0 0 "$cfg"; run 4096 0 "$cfg"; run 0 4096 "$cfg"
done
