# fused sweep at several D per library variant (default = product library); usage: ab_wide_variants.sh ["D Np B" ...]
cfgs=("$@"); [ ${#cfgs[@]} -gt 0 ] || cfgs=("384 201 1" "640 101 1" "1000 81 1" "1536 41 1")
for lib in default $(ls vgpa_amd/lib/variants 2>/dev/null | sed 's/libvgpa_hip_//; s/\.so//'); do
  unset VGPA_LIB; [ $lib = default ] || export VGPA_LIB=$PWD/vgpa_amd/lib/variants/libvgpa_hip_$lib.so
  for c in "${cfgs[@]}"; do echo "LIB=$lib : $c"; python tools/bench_large_d_sweep.py $c; done
done
