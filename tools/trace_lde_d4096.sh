export TMPDIR=/tmp
export VGPA_HEAD=$(cat vgpa_amd/_tree.txt 2>/dev/null)
rm -rf gpurun_out/tr_c5
rocprofv3 --kernel-trace -d gpurun_out/tr_c5 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-problem --no-config2 --no-config4 > gpurun_out/tr_c5.json 2> gpurun_out/tr_c5.err
DB=$(find gpurun_out/tr_c5 -name "*results.db" | head -1)
{ echo "# tree $VGPA_HEAD: last energy batch of bench.py's config5 block (D = 4096, Np = 64, one rank)"; python3 tools/trace_lde.py $DB; } > gpurun_out/r05g_lde_trace_D4096.txt
rm -rf gpurun_out/tr_c5
cat gpurun_out/r05g_lde_trace_D4096.txt
