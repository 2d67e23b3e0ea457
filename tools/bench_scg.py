"""Time SCG iterations at BASELINE config 3 (L96 D=40 RK4 Np=1001): host SCG over the GPU objective vs DeviceSCG
(vectors resident in HBM), single problem and a lock-step batch.  Prints one JSON line per case."""
import json
import sys
import time

import numpy as np

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import vgpa_amd as va                       # noqa: E402
from helpers import build_problem           # noqa: E402

IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = build_problem("L96", "RK4", 10.0, 0.01, 40)
v = p["vgp"]
x0 = v.initialization()
opts = {"max_it": IT, "x_tol": 0.0, "f_tol": 0.0}
v.free_energy(x0)

t0 = time.perf_counter()
host = va.SCG(v.free_energy, v.gradient, dict(opts))
_, f_h = host(x0.copy())
t_host = time.perf_counter() - t0
print(json.dumps({"case": "host SCG + GPU objective", "B": 1, "iterations": IT, "seconds": t_host,
                  "it_per_s": IT / t_host, "f_final": f_h, "f_eval": host.statistics["f_eval"]}))

t0 = time.perf_counter()
dev = v.device_scg(dict(opts))
_, f_d = dev(x0.copy())
t_dev = time.perf_counter() - t0
print(json.dumps({"case": "DeviceSCG", "B": 1, "iterations": IT, "seconds": t_dev, "it_per_s": IT / t_dev,
                  "f_final": f_d, "f_eval": dev.statistics["f_eval"], "rel_diff_vs_host": abs(f_d - f_h) / abs(f_h)}))

for B in (16, 128):
    args = (p["model"], p["m0"], p["s0"], p["fwd"], p["bwd"], p["lik"], p["kl0"], p["obs_y"], p["obs_t"])
    vb = va.VarGP(*args, batch=B)
    rng = np.random.default_rng(0)
    xs = x0[None, :] + 0.01 * rng.standard_normal((B, x0.size))
    xs[0] = x0
    run = vb.device_scg(dict(opts))
    t0 = time.perf_counter()
    _, fb = run(xs)
    t_b = time.perf_counter() - t0
    print(json.dumps({"case": "DeviceSCG lock-step batch", "B": B, "iterations": IT, "seconds": t_b,
                      "problem_it_per_s": B * IT / t_b, "f_final_problem0": float(fb[0]),
                      "rel_diff_problem0_vs_single": abs(fb[0] - f_d) / abs(f_d)}))
    del vb, run
