set -e
python -m pytest tests/test_large_d.py tests/test_gpu_edge_cases.py -q -m gpu -x 2>&1 | tail -3
for mode in 0 512; do
  for cfg in "96 401 1" "128 401 1" "256 401 1" "512 201 1" "128 401 8" "1024 41 1"; do
    echo "VGPA_STAGE_FUSED=$mode $cfg"
    VGPA_STAGE_FUSED=$mode python tools/bench_large_d_sweep.py $cfg
  done
done
VGPA_STAGE_FUSED=1024 python tools/bench_large_d_sweep.py 1024 41 1
