#!/usr/bin/env python3
"""The reference's whole workflow at BASELINE configs[2] (sim_params_L40D.json-style parameters, RK4): Simulation.setup ->
run (SCG, 500 iterations, device-resident vectors) -> save.  Prints one JSON line with the wall time of the optimisation."""
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

params = {"Model": "L96", "Ode-method": "RK4", "Random-Seed": 31415926535,
          "Time-window": {"t0": 0.0, "tf": 10.0, "dt": 0.01}, "Noise": {"sys": [4.0] * 40, "obs": 1.0},
          "Observations": {"density": 8, "operator": None}, "Drift": {"theta": 8.0}, "Prior": {"mu0": 1.0, "tau0": 0.5}}
max_it = int(sys.argv[1]) if len(sys.argv) > 1 else 500
os.chdir(os.environ.get("TMPDIR", "/tmp"))
sim = va.Simulation("L40D_rk4")
with contextlib.redirect_stdout(io.StringIO()):
    sim.setup(params, None)
    t0 = time.perf_counter()
    sim.run(options={"max_it": max_it, "x_tol": 1.0e-6, "f_tol": 1.0e-8, "display": False}, device_resident=True)
    t_run = time.perf_counter() - t0
    sim.save()
out = va.load_results("L40D_rk4.h5")
print(json.dumps({"workflow": "Simulation.setup -> run(DeviceSCG, max_it=%d) -> save, Lorenz96 D=40 RK4 Np=1001" % max_it,
                  "seconds_optimisation": t_run, "F_final": float(out["fx"][0]),
                  "file_keys": sorted(out), "mt_shape": list(out["mt"].shape),
                  "rmse_mean_vs_true_path": float(np.sqrt(np.mean((out["mt"] - sim.m_data["model"].sample_path) ** 2)))}))
