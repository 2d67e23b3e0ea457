#!/bin/bash
# The three implementations of a Runge-Kutta stage above D = 64 (large_d.hip: GEMM + k_stage_sym, k_stage_prod, k_stage_wide) on ONE box:
#   (1) the fused sweep (tools/bench_large_d_sweep.py) per size and version -> gpurun_out/<tag>_stage_versions_ab.txt
#   (2) rocprofv3 kernel stats + matrix-pipe counters at D = 1024 for the two-kernel stage and for k_stage_wide
#       -> gpurun_out/<tag>_stage_<version>_D1024_{kernel_stats,mfma_counters}.csv
#   (3) library variants under vgpa_amd/lib/variants (tools/build_variant.sh <name> "<flags>" large_d.hip), if any, at a few sizes.
# usage: bash tools/stamp_tree.sh && gpurun -- 'bash tools/profile_stage_versions.sh r05'
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}
TAG=${1:-r05}
export TMPDIR=/tmp
out=gpurun_out/${TAG}_stage_versions_ab.txt
echo "# tree $VGPA_HEAD; fused sweep of one Lorenz-96 problem per context (B = problems per context), ms per sweep; version = VGPA_STAGE_FUSED / VGPA_STAGE_WIDE" > $out
echo "# two-kernel = 0/0 (GEMM + k_stage_sym), prod = 4096/0 (k_stage_prod), wide = 0/4096 (k_stage_wide), default = the library's rule" >> $out
run() { VGPA_STAGE_FUSED=$2 VGPA_STAGE_WIDE=$3 python3 tools/bench_large_d_sweep.py $4 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['phase_ms']
print('%-10s D %4d Np %4d B %d  sweep %7.2f ms  fwd %6.2f bwd %6.2f energy %6.2f grad %5.2f  fwd %5.1f TF  F %.10g' % ('$1', d['D'], d['Np'], d['batch'], d['ms_per_sweep'], p['fwd_ms'], p['bwd_ms'], p['energy_ms'], p['grad_ms'], d['fwd_tflops'], d['F']))" >> $out; }
for cfg in "72 401 1" "96 401 1" "128 401 1" "200 401 1" "256 401 1" "384 201 1" "512 201 1" "128 401 8" "640 101 1" "768 101 1" "1000 81 1" "1024 81 1" "1536 41 1" "2048 21 1"; do
  run two-kernel 0 0 "$cfg"; run prod 4096 0 "$cfg"; run wide 0 4096 "$cfg"
  python3 tools/bench_large_d_sweep.py $cfg | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s D %4d Np %4d B %d  sweep %7.2f ms' % ('default', d['D'], d['Np'], d['batch'], d['ms_per_sweep']))" >> $out
done
for v in two-kernel wide; do
  if [ $v = wide ]; then export VGPA_STAGE_WIDE=4096; else export VGPA_STAGE_WIDE=0; fi
  export VGPA_STAGE_FUSED=0
  rm -rf gpurun_out/ps_stats gpurun_out/ps_mfma
  rocprofv3 --kernel-trace --stats -d gpurun_out/ps_stats -- python3 tools/bench_large_d_sweep.py 1024 41 > gpurun_out/ps.json 2> gpurun_out/ps.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/ps_mfma -- python3 tools/bench_large_d_sweep.py 1024 41 > gpurun_out/ps_mfma.json 2> gpurun_out/ps_mfma.err
  S=$(find gpurun_out/ps_stats -name "*results.db" | head -1); M=$(find gpurun_out/ps_mfma -name "*results.db" | head -1)
  python3 tools/rocprof_db.py stats $S gpurun_out/${TAG}_stage_${v}_D1024_kernel_stats.csv
  python3 tools/rocprof_db.py counters $M gpurun_out/${TAG}_stage_${v}_D1024_mfma_counters.csv "bench_large_d_sweep.py 1024 41, stage = $v"
  rm -rf gpurun_out/ps_stats gpurun_out/ps_mfma
done
unset VGPA_STAGE_WIDE VGPA_STAGE_FUSED
for lib in $(ls vgpa_amd/lib/variants 2>/dev/null | sed 's/libvgpa_hip_//; s/\.so//'); do
  for cfg in "128 401 1" "384 201 1" "1000 81 1" "1536 41 1"; do
    VGPA_LIB=$PWD/vgpa_amd/lib/variants/libvgpa_hip_$lib.so python3 tools/bench_large_d_sweep.py $cfg | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s D %4d Np %4d B %d  sweep %7.2f ms' % ('lib:$lib', d['D'], d['Np'], d['batch'], d['ms_per_sweep']))" >> $out
  done
done
cat $out
