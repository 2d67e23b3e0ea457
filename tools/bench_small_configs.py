#!/usr/bin/env python3
"""BASELINE configs[0] (OU, Euler) and configs[1] (Lorenz-63, RK4), t in [0, 10], dt = 0.01 (Np = 1001): sweeps/s of the
fused free-energy + gradient evaluation for one problem and for a batch of independent problems.  One JSON line each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vgpa_amd as va                      # noqa: E402
from helpers import build_problem          # noqa: E402


def run(model, method, batch, reps=20, flags=0):
    p = build_problem(model, method, 10.0, 0.01, None)
    v = p["vgp"]
    x0 = v.initialization()
    d = v.dim_d
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    ctx = va.Context(model, method, d, v.dim_n, 0.01, sigma=p["model"].sigma, theta=p["model"].theta, m0=p["m0"],
                     s0=p["s0"], obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=batch, flags=flags)
    noise = 0.01 * np.random.default_rng(0).standard_normal((min(batch, 61), x0.size))     # 61 distinct problems, repeated
    xb = np.empty((batch, x0.size))
    for i0 in range(0, batch, noise.shape[0]):
        k = min(noise.shape[0], batch - i0)
        xb[i0:i0 + k] = x0 + noise[:k]
    xd, gd = ctx.alloc(batch * x0.size), ctx.alloc(batch * x0.size)
    xd.upload(xb)
    for _ in range(3):
        ctx.sweep_enqueue(xd, gd); ctx.fetch_f()
    t0 = time.perf_counter()               # the timed loop runs WITHOUT the phase events (their queries cost ~0.4 ms per step: a quarter of an OU step)
    for _ in range(reps):
        ctx.sweep_enqueue(xd, gd); f = ctx.fetch_f()
    dt = (time.perf_counter() - t0) / reps
    ctx.profile_begin()
    for _ in range(reps):
        ctx.sweep_enqueue(xd, gd); ctx.fetch_f()
    pr = ctx.profile_end()
    ctx.close()
    n = int(v.dim_n)
    alg = 8.0 * n * (5 * d * d + 6 * d)            # algorithmic bytes of one fused sweep
    return {"model": model, "method": method, "D": d, "Np": n, "batch": batch, "flags": flags, "ms_per_step": 1e3 * dt,
            "sweeps_per_s": batch / dt, "alg_bytes_per_sweep": alg, "GBs": alg * batch / dt / 1e9, "frac_of_8TBs": alg * batch / dt / 8e12,
            "F0": float(np.atleast_1d(f)[0]),
            "phase_ms": {k: pr[k] / reps for k in ("fwd_ms", "energy_ms", "bwd_ms", "grad_ms")}}


if __name__ == "__main__":
    # usage: bench_small_configs.py [batch,batch,...] [MODEL] [reps] [fused-only]
    from vgpa_amd._lib import FLAG_MATERIALIZE
    batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 65536]
    if len(sys.argv) > 2:          # one model, the fused path only: what a counter pass wants
        m = sys.argv[2].upper()
        reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
        for batch in batches:
            print(json.dumps(run(m, "Euler" if m == "OU" else "RK4", batch, reps=reps)), flush=True)
        sys.exit(0)
    for model, method in (("OU", "Euler"), ("L63", "RK4")):
        for batch in batches:
            for flags in ((0, FLAG_MATERIALIZE) if batch >= 512 or model == "OU" else (0,)):
                print(json.dumps(run(model, method, batch, flags=flags)), flush=True)
