set -e
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}      # the commit of the measured tree (tools/stamp_tree.sh), into every summary header
TAG=${1:-r04}; MODEL=${2:-L63}; BATCH=${3:-65536}   # usage: bash tools/profile_small.sh [tag] [model] [batch]: writes gpurun_out/<tag>_*
export TMPDIR=/tmp
rm -rf gpurun_out/ps_stats gpurun_out/ps_fetch gpurun_out/ps_write
rocprofv3 --kernel-trace --stats -d gpurun_out/ps_stats -- python3 tools/bench_small_configs.py $BATCH $MODEL 10 > gpurun_out/${TAG}_small_${MODEL}_B${BATCH}.json 2> gpurun_out/ps_stats.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/ps_fetch -- python3 tools/bench_small_configs.py $BATCH $MODEL 2 > gpurun_out/ps_fetch.json 2> gpurun_out/ps_fetch.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/ps_write -- python3 tools/bench_small_configs.py $BATCH $MODEL 2 > gpurun_out/ps_write.json 2> gpurun_out/ps_write.err
S=$(find gpurun_out/ps_stats -name "*results.db" | head -1); F=$(find gpurun_out/ps_fetch -name "*results.db" | head -1); W=$(find gpurun_out/ps_write -name "*results.db" | head -1)
python3 tools/rocprof_db.py stats $S gpurun_out/${TAG}_small_${MODEL}_B${BATCH}_kernel_stats.csv
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python3 tools/rocprof_db.py traffic_all $F $W gpurun_out/${TAG}_small_${MODEL}_B${BATCH}_pmc_fetch_write.csv "python tools/bench_small_configs.py $BATCH $MODEL" gpurun_out/pmc_traffic.json ${MODEL}_sweep_B${BATCH} "k_fwd_lane|k_obs|k_sweep_lane"
rm -rf gpurun_out/ps_stats gpurun_out/ps_fetch gpurun_out/ps_write
