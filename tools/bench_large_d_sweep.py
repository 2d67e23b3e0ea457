#!/usr/bin/env python3
"""Fused sweep (free energy + gradient) of a Lorenz-96 problem at large D on one GPU, through the C ABI with
device-resident x / gradient.  Prints one JSON line: ms per sweep, per phase, fp64 TFLOP/s of the recursions."""
import os
import sys
import json
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import vgpa_amd as va
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 41
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # problems per context (grid.z of the per-stage kernels)
    rng = np.random.default_rng(1)
    m0 = 8.0 + rng.standard_normal(d)
    obs_t = np.arange(4, n - 1, 12, dtype=np.int64)
    obs_y = 8.0 + rng.standard_normal((obs_t.size, d))
    ctx = va.Context("L96", "rk4", d, n, 0.01, sigma=4.0 * np.eye(d), theta=[8.0], m0=m0, s0=0.2 * np.eye(d), obs_t=obs_t,
                     obs_y=obs_y, obs_noise=np.eye(d), e0=0.0, batch=nb)
    a = 8.0 * np.eye(d)[None] + 0.05 * np.random.default_rng(2).standard_normal((n, d, d)) / np.sqrt(d)
    b = 8.0 * m0[None] + np.random.default_rng(3).standard_normal((n, d))
    x = np.concatenate((a.ravel(), b.ravel()))
    x = np.stack([x + 0.01 * p * np.sin(np.arange(x.size)) for p in range(nb)]) if nb > 1 else x
    xd, gd = ctx.alloc(x.size), ctx.alloc(x.size)
    xd.upload(x)
    ctx.sweep_dev(xd, gd)
    ctx.profile_begin()
    reps = 5
    ctx.sweep_dev(xd, gd)
    t0 = time.perf_counter()
    for _ in range(reps):
        f = ctx.sweep_dev(xd, gd)
    dt = (time.perf_counter() - t0) / reps
    pr = ctx.profile_end()
    flop_rec = nb * (n - 1) * 4 * 2.0 * d ** 3
    f = float(np.atleast_1d(f)[0])
    out = {"D": d, "Np": n, "batch": nb, "F": f, "ms_per_sweep": 1e3 * dt,
           "phase_ms": {k: pr[k] / reps for k in ("fwd_ms", "energy_ms", "bwd_ms", "grad_ms")},
           "fwd_tflops": flop_rec / (pr["fwd_ms"] / reps) / 1e9, "bwd_tflops": flop_rec / (pr["bwd_ms"] / reps) / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
