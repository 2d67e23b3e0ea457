#!/bin/bash
# Clock and package power with ONE kernel of the fused sweep held on the chip (VGPA_DIAG_REPEAT launches that phase 40 times per
# sweep; the results -- and the bench's anchor check -- do not change).  Writes gpurun_out/<tag>_power_per_kernel.txt
TAG=${1:-r03}
out=gpurun_out/${TAG}_power_per_kernel.txt
: > $out
for ph in fwd energy bwd grad; do
  VGPA_DIAG_REPEAT=$ph:40 python bench.py --steps 40 --warmup 1 --no-cpu-baseline --no-config5 --no-single-problem > gpurun_out/ppk_$ph.json 2> gpurun_out/ppk_$ph.err &
  pid=$!
  sleep 10
  echo "== $ph" >> $out
  for i in 1 2 3 4 5 6; do
    if ! kill -0 $pid 2>/dev/null; then break; fi
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | paste - - | awk '{print "   " $7, $NF " W"}' >> $out; sleep 0.5
  done
  wait $pid
  python3 - $ph >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ppk_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = {"fwd": "fwd", "energy": "energy+obs", "bwd": "bwd", "grad": "reduce+grad"}[sys.argv[1]]
print("   %s: %.3f ms per launch (40 launches per sweep), parity F %.1e" % (sys.argv[1], d["phase_ms_per_step"][k] / 40.0, d["parity_check_rel_err_F"]))
PY
done
cat $out
