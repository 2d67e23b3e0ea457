#!/bin/bash
# Same-box A/B of kernel build variants (boxes of the pool differ by 2-4 %): each variant = extra compiler flags (-D macros of the
# steppers, the energy kernel or the gradient assembly), rebuilt on the box, benched twice.
#   tools/ab_steppers.sh "name1:flags1" "name2:flags2" ...
set -u
out=gpurun_out/ab_$(date +%H%M%S).txt
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  touch ${AB_FILES:-vgpa_amd/csrc/ode_sym_impl.h vgpa_amd/csrc/energy.hip vgpa_amd/csrc/assemble.hip}      # AB_FILES: only the sources the flags concern
  VGPA_EXTRA_CFLAGS="$flags" python -m vgpa_amd.build > gpurun_out/ab_build_$name.log 2>&1 || { echo "$name: build failed" | tee -a $out; continue; }
  for rep in 1 2; do
    python bench.py --steps ${AB_STEPS:-10} --warmup 3 ${AB_BENCH_ARGS:---no-single-problem} --no-cpu-baseline --no-config5 --no-config4 --no-config2 > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
    python - "$name" "$flags" <<PY | tee -a $out
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
p = d["phase_ms_per_step"]
sp = d.get("single_problem") or {}
print("%-14s fwd %.3f bwd %.3f energy %.3f grad %.3f | %.0f sweeps/s | one problem fwd %s bwd %s | %s" % (sys.argv[1], p["fwd"], p["bwd"], p["energy+obs"], p["reduce+grad"], d["value"], sp.get("fwd_ms"), sp.get("bwd_ms"), sys.argv[2]))
PY
  done
done
