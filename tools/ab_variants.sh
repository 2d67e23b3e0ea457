# usage: ab.sh name1 name2 ...  ("default" = product lib)
for rep in 1 2; do for n in "$@"; do
  if [ $n = default ]; then L=""; else L="VGPA_LIB=$PWD/vgpa_amd/lib/variants/libvgpa_hip_$n.so"; fi
  env $L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-config2 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python3 - $n <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
p = d["phase_ms_per_step"]; sp = d.get("single_problem") or {}
print("%-10s fwd %.3f en %.3f bwd %.3f grad %.3f | step %.3f ms %.0f sweeps/s | one problem %.3f ms (fwd %.3f bwd %.3f) | chk %.1e %.1e" % (sys.argv[1], p["fwd"], p["energy+obs"], p["bwd"], p["reduce+grad"], d["ms_per_step"], d["value"], sp.get("ms_per_sweep", 0), sp.get("fwd_ms", 0), sp.get("bwd_ms", 0), d["parity_check_rel_err_F"], d["parity_check_rel_err_grad_norm"]))
PY
done; done
