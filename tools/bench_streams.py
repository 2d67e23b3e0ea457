#!/usr/bin/env python3
"""Experiment: the default bench workload (Lorenz-96 D = 40, RK4, Np = 1001) split over K contexts = K HIP streams, so that
the HBM-bound kernels of one sub-batch (energy terms, gradient assembly) can run beside the matrix-core-bound steppers of
another.  Prints one JSON line per (total batch, K, kernel family).

    python tools/bench_streams.py [total_batch=512] [steps=6]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import build_problem
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    d, n_pts, dt = 40, 1001, 0.01
    p = build_problem("L96", "RK4", (n_pts - 1) * dt, dt, d)
    x0 = p["vgp"].initialization()
    import torch  # noqa: F401  (one HIP runtime)
    import vgpa_amd as va
    from vgpa_amd._lib import FLAG_SYM_UNITS
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    len_x = x0.size
    for flags, fam in ((0, "auto"), (FLAG_SYM_UNITS, "sym")):
        for k in (1, 2, 4):
            b = total // k
            ctxs, bufs = [], []
            for c in range(k):
                ctx = va.Context("L96", "RK4", d, n_pts, dt, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"],
                                 obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=b, flags=flags)
                xb = np.stack([x0 + 0.05 * np.random.default_rng(1000 + c * b + i).standard_normal(len_x) for i in range(b)])
                xd, gd = ctx.alloc(b * len_x), ctx.alloc(b * len_x)
                xd.upload(xb)
                ctxs.append(ctx)
                bufs.append((xd, gd))
            for _ in range(2):
                for ctx, (xd, gd) in zip(ctxs, bufs):
                    ctx.sweep_enqueue(xd, gd)
                for ctx in ctxs:
                    ctx.fetch_f()
            t0 = time.perf_counter()
            for _ in range(steps):
                for ctx, (xd, gd) in zip(ctxs, bufs):
                    ctx.sweep_enqueue(xd, gd)
                for ctx in ctxs:
                    ctx.fetch_f()
            el = time.perf_counter() - t0
            print(json.dumps({"total_batch": total, "contexts": k, "batch_per_context": b, "kernels": fam,
                              "ms_per_step": 1e3 * el / steps, "sweeps_per_s": total * steps / el}), flush=True)
            for ctx in ctxs:
                ctx.close()


if __name__ == "__main__":
    main()
