#!/bin/bash
# A/B on one box: four waves per problem (two workgroups per CU) against eight (VGPA_SYM_WAVES), single problem and batches
for W in 4 8; do
  for B in 1 256 512; do
    VGPA_SYM_WAVES=$W python - $W $B <<'PY'
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import vgpa_amd as va
from helpers import build_problem
W, B = int(sys.argv[1]), int(sys.argv[2])
p = build_problem("L96", "RK4", 10.0, 0.01, 40)
x0 = p["vgp"].initialization()
e0 = float(p["kl0"](p["m0"], p["s0"]))
ctx = va.Context("L96", "RK4", 40, 1001, 0.01, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"], obs_t=p["obs_t"], obs_y=p["obs_y"],
                 obs_noise=p["obs_noise"], e0=e0, batch=B)
xb = np.stack([x0 + 0.05 * np.random.default_rng(i).standard_normal(x0.size) for i in range(min(B, 8))])
xb = np.tile(xb, ((B + 7) // 8, 1))[:B]
xd, gd = ctx.alloc(B * x0.size), ctx.alloc(B * x0.size)
xd.upload(xb)
for _ in range(2):
    ctx.sweep_enqueue(xd, gd); ctx.fetch_f()
ctx.profile_begin()
reps = 6
t0 = time.perf_counter()
for _ in range(reps):
    ctx.sweep_enqueue(xd, gd); f = ctx.fetch_f()
dt = (time.perf_counter() - t0) / reps
pr = ctx.profile_end()
print(json.dumps({"waves": W, "batch": B, "ms_per_sweep": round(1e3 * dt, 3), "fwd_ms": round(pr["fwd_ms"] / reps, 3), "bwd_ms": round(pr["bwd_ms"] / reps, 3),
                  "F0": float(np.atleast_1d(f)[0])}), flush=True)
PY
  done
done
