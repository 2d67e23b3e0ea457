#!/usr/bin/env python3
"""The reference's whole workflow at BASELINE configs[2] (sim_params_L40D.json-style parameters, RK4): seeded inputs ->
DeviceSCG (500 iterations, device-resident vectors) -> save_results.  Prints one JSON line with the wall time of the optimisation."""
import io
import json
import os
import sys
import time
import contextlib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import vgpa_amd as va       # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_problem   # noqa: E402  (the seeded reference inputs of configs[2])

max_it = int(sys.argv[1]) if len(sys.argv) > 1 else 500
os.chdir(os.environ.get("TMPDIR", "/tmp"))
with contextlib.redirect_stdout(io.StringIO()):
    p = build_problem("L96", "RK4", 10.0, 0.01, 40)
    v = p["vgp"]
    opt = v.device_scg({"max_it": max_it, "x_tol": 1.0e-6, "f_tol": 1.0e-8, "display": False})
    x0 = v.initialization()
    t0 = time.perf_counter()
    x, fx = opt(x0)
    t_run = time.perf_counter() - t0
    path, _ = va.save_results("L40D_rk4", v, x, fx)
out = va.load_results(path)
print(json.dumps({"workflow": "build inputs -> DeviceSCG(max_it=%d) -> save_results, Lorenz96 D=40 RK4 Np=1001" % max_it,
                  "seconds_optimisation": t_run, "F_final": float(out["fx"][0]),
                  "file_keys": sorted(out), "mt_shape": list(out["mt"].shape),
                  "rmse_mean_vs_true_path": float(np.sqrt(np.mean((out["mt"] - p["model"].sample_path) ** 2)))}))
