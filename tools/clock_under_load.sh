#!/bin/bash
# Shader / memory clocks and package power while the headline bench runs (rocm-smi samples every 0.5 s): what the "peak" of the
# fp64 matrix pipe (78.6 TFLOP/s at 2.4 GHz) is worth on a box under this load.  Writes gpurun_out/<tag>_clock_under_load.txt
TAG=${1:-r03}
out=gpurun_out/${TAG}_clock_under_load.txt
python bench.py --batch ${BATCH:-512} --steps ${STEPS:-900} --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-config2 --no-single-problem > gpurun_out/${TAG}_clock_bench.json 2> gpurun_out/${TAG}_clock_bench.err &
pid=$!
sleep 9      # import + context + warm-up
: > $out
for i in $(seq 1 30); do
  if ! kill -0 $pid 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $out
  echo "--" >> $out
  sleep 0.5
done
wait $pid
echo "idle:" >> $out
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" >> $out
cat $out
