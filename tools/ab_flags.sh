#!/bin/bash
# Same-box A/B of build variants of ONE source file (the other objects are reused): tools/ab_flags.sh <file under vgpa_amd/csrc> "name:flags" ...
set -u
src=$1; shift
out=gpurun_out/ab_$(date +%H%M%S).txt
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  touch vgpa_amd/csrc/$src
  VGPA_EXTRA_CFLAGS="$flags" python -m vgpa_amd.build > gpurun_out/ab_build_$name.log 2>&1 || { echo "$name: build failed" | tee -a $out; continue; }
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-single-problem > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
    python - "$name" "$flags" <<PY | tee -a $out
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
p = d["phase_ms_per_step"]
print("%-14s fwd %.3f bwd %.3f energy %.3f grad %.3f | %.0f sweeps/s | %s" % (sys.argv[1], p["fwd"], p["bwd"], p["energy+obs"], p["reduce+grad"], d["value"], sys.argv[2]))
PY
  done
done
