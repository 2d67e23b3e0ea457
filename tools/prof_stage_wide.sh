# kernel times of the fused sweep at D = 1024 per stage implementation and library variant (tools/build_variant.sh <name> "<flags>" large_d.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VGPA_ALLOW_DIAGNOSTIC=1
for lib in default $(ls $R/vgpa_amd/lib/variants 2>/dev/null | sed 's/libvgpa_hip_//; s/\.so//'); do
  for mode in two-kernel wide; do
    unset VGPA_LIB
    [ $lib = default ] || export VGPA_LIB=$R/vgpa_amd/lib/variants/libvgpa_hip_$lib.so
    if [ $mode = wide ]; then export VGPA_STAGE_WIDE=2048; else export VGPA_STAGE_WIDE=0; fi
    case $lib in w*) [ $mode = wide ] || continue;; g*) [ $mode = two-kernel ] || continue;; esac
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wide_${lib}_$mode -o p -- python $R/tools/bench_large_d_sweep.py 1024 21 1 > $R/gpurun_out/prof_wide_${lib}_$mode.log 2>&1 || echo "$lib $mode: failed"
  done
done
