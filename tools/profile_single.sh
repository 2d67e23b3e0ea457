# rocprofv3 kernel stats of ONE problem's sweeps (the helper-wave steppers): tools/profile_single.sh [tag]
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}      # the commit of the measured tree (tools/stamp_tree.sh), into every summary header
set -e
TAG=${1:-r04z}
export TMPDIR=/tmp
rm -rf gpurun_out/pss
rocprofv3 --kernel-trace --stats -d gpurun_out/pss -- python3 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-config2 --no-single-problem > gpurun_out/${TAG}_bench_B1.json 2> gpurun_out/pss.err
S=$(find gpurun_out/pss -name "*results.db" | head -1)
python3 tools/rocprof_db.py stats $S gpurun_out/${TAG}_bench_B1_kernel_stats.csv
rm -rf gpurun_out/pss
