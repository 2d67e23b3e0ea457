#!/bin/bash
# The fp64 matrix pipe's SUSTAINED rate on the whole chip under the power cap, with clock and package power sampled beside it
# (tools/ubench/f64_sustained.hip).  Writes gpurun_out/<tag>_sustained_fp64_peak.txt
TAG=${1:-r03}
out=gpurun_out/${TAG}_sustained_fp64_peak.txt
hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_sustained.hip -o gpurun_out/f64_sustained || exit 1
: > $out
for cfg in "0 1 rand" "0 2 rand" "0 2 rand8" "0 1 rand8" "1 2 rand" "1 2 rand8" "0 2 zero"; do
  set -- $cfg
  gpurun_out/f64_sustained $1 $2 5 $3 > gpurun_out/f64_sustained.log &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | paste - - | awk '{print "   " $7, $NF " W"}' >> $out; sleep 0.5; done
  wait $pid
  tail -1 gpurun_out/f64_sustained.log >> $out
done
rm -f gpurun_out/f64_sustained gpurun_out/f64_sustained.log
cat $out
