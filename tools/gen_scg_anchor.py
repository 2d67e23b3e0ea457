#!/usr/bin/env python3
"""Runs the REFERENCE's own optimiser on the REFERENCE's own objective (imported from /root/reference through
tools/gen_golden.py's identity-njit shim; build container only) and prints the trace a golden file holds:

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_scg_anchor.py L96 RK4 10.0 3    > tests/golden/scg_trace_config3.json   (2 min)
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_scg_anchor.py L96 RK4 10.0 500  > tests/golden/scg_full_config3.json    (9 min)
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_scg_anchor.py OU Euler 10.0 500 > tests/golden/scg_full_config1.json
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_scg_anchor.py L63 RK4 10.0 500  > tests/golden/scg_full_config2.json

Arguments: model, stepper, t_f, max_it (SCG stops earlier by its own criteria).  Only numbers are written."""
import io
import json
import os
import sys
import time
import contextlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                                  # noqa: E402  (installs the shim, imports the reference)
from src.numerics.optim_scg import SCG                   # noqa: E402


def main():
    name, method, tf, max_it = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    c = gg.build(name, method, tf)
    v = c["vgp"]
    opt = SCG(v.free_energy, v.gradient, {"max_it": max_it, "x_tol": 1e-6, "f_tol": 1e-8, "display": False})
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        x, fx = opt(c["x0"].copy())
    st = opt.stats
    n = int(st["MaxIt"])
    print(json.dumps({"model": name, "method": method, "tf": tf, "max_it": max_it, "MaxIt_stat": n, "f_eval": float(st["f_eval"]),
                      "seconds": time.perf_counter() - t0, "fx_trace": [float(a) for a in st["fx"][:n]],
                      "beta_trace": [float(a) for a in st["beta"][:n]], "f_final": float(fx),
                      "x_norm": float(np.linalg.norm(x))}))


if __name__ == "__main__":
    main()
