set -e
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}      # the commit of the measured tree (tools/stamp_tree.sh), into every summary header
# usage: bash tools/profile_large_d.sh [tag]: rocprofv3 kernel stats + matrix-pipe counters of (a) the resident fused sweep at D = 1024, Np = 41
# (tools/bench_large_d_sweep.py) and (b) the one-rank D = 4096 fused sweep through the row-sharded driver (tools/bench_config5.py --sweep).
# Counters in their own passes (no trace domains beside --pmc).  Writes gpurun_out/<tag>_large_d_*.
TAG=${1:-r04}
export TMPDIR=/tmp
run_case() {   # name, command...
  NAME=$1; shift
  rm -rf gpurun_out/pl_stats gpurun_out/pl_mfma gpurun_out/pl_busy
  rocprofv3 --kernel-trace --stats -d gpurun_out/pl_stats -- "$@" > gpurun_out/${TAG}_large_d_${NAME}.json 2> gpurun_out/pl_stats.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/pl_mfma -- "$@" > gpurun_out/pl_mfma.json 2> gpurun_out/pl_mfma.err || \
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pl_mfma -- "$@" > gpurun_out/pl_mfma.json 2> gpurun_out/pl_mfma.err
  S=$(find gpurun_out/pl_stats -name "*results.db" | head -1); M=$(find gpurun_out/pl_mfma -name "*results.db" | head -1)
  python3 tools/rocprof_db.py stats $S gpurun_out/${TAG}_large_d_${NAME}_kernel_stats.csv
  python3 tools/rocprof_db.py counters $M gpurun_out/${TAG}_large_d_${NAME}_mfma_counters.csv "$*"
  rm -rf gpurun_out/pl_stats gpurun_out/pl_mfma
}
run_case D1024_Np41_sweep python3 tools/bench_large_d_sweep.py 1024 41
run_case D4096_Np16_one_rank_sweep python3 tools/bench_config5.py --sweep --dim 4096 --np 16 --reps 2
