# fused sweep at several D per library variant (default = product library)
for lib in default $(ls vgpa_amd/lib/variants 2>/dev/null | sed 's/libvgpa_hip_//; s/\.so//'); do
  unset VGPA_LIB; [ $lib = default ] || export VGPA_LIB=$PWD/vgpa_amd/lib/variants/libvgpa_hip_$lib.so
  for c in "384 201 1" "640 101 1" "1000 81 1" "1536 41 1"; do echo "LIB=$lib : $c"; VGPA_STAGE_WIDE=4096 python tools/bench_large_d_sweep.py $c; done
done
