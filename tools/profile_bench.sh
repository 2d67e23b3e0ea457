set -e
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}      # the commit of the measured tree (tools/stamp_tree.sh), into every summary header
TAG=${1:-r03}          # usage: bash tools/profile_bench.sh [tag]: writes gpurun_out/<tag>_* (copy what is to be judged into profiles/)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-problem --no-config5 --no-config4 --no-config2 > gpurun_out/${TAG}_bench_B512_batched_only.json 2> gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-problem --no-config5 --no-config4 --no-config2 > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-problem --no-config5 --no-config4 --no-config2 > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err
S=$(find gpurun_out/prof_stats -name "*results.db" | head -1); F=$(find gpurun_out/prof_fetch -name "*results.db" | head -1); W=$(find gpurun_out/prof_write -name "*results.db" | head -1)
python3 tools/rocprof_db.py stats $S gpurun_out/${TAG}_bench_B512_batched_only_kernel_stats.csv
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python3 tools/rocprof_db.py traffic $F $W 512 40 1001 gpurun_out/${TAG}_pmc_fetch_write_B512.csv gpurun_out/pmc_traffic.json
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
