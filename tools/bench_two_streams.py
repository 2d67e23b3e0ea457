#!/usr/bin/env python3
"""BASELINE configs[2] (Lorenz-96, D = 40, RK4, Np = 1001): the batch as ONE context against the same problems split over 2 / 4
contexts, each on its own stream, their sweeps enqueued back to back without a host sync in between -- do kernels of different
phases (matrix-pipe-bound steppers, latency-bound energy terms, HBM-bound gradient assembly) of different sub-batches overlap?
    python tools/bench_two_streams.py [batch] [sweeps]
One JSON line per split."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from helpers import build_problem
    import vgpa_amd as va
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    d, n_pts, dt = 40, 1001, 0.01
    p = build_problem("L96", "rk4", (n_pts - 1) * dt, dt, d)
    x0 = p["vgp"].initialization()
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    xb = np.stack([x0 + 0.05 * np.random.default_rng(1000 + i).standard_normal(x0.size) for i in range(B)])

    def make(b, lo):
        ctx = va.Context("L96", "rk4", d, n_pts, dt, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"],
                         obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=b)
        xd, gd = ctx.alloc(b * x0.size), ctx.alloc(b * x0.size)
        xd.upload(xb[lo:lo + b])
        return ctx, xd, gd

    f_ref = None
    for parts in (1, 2, 4):
        b = B // parts
        cs = [make(b, i * b) for i in range(parts)]
        for _ in range(2):
            for ctx, xd, gd in cs:
                ctx.sweep_enqueue(xd, gd)
            fs = [np.atleast_1d(ctx.fetch_f()) for ctx, _, _ in cs]
        f = np.concatenate(fs)
        if f_ref is None:
            f_ref = f
        for stagger in ((False,) if parts == 1 else (False, True)):
            t0 = time.perf_counter()
            if stagger:      # sub-batch i starts i phases late: its first sweeps are enqueued behind those of the ones before it
                for k in range(K + parts - 1):
                    for i, (ctx, xd, gd) in enumerate(cs):
                        if 0 <= k - i < K:
                            ctx.sweep_enqueue(xd, gd)
            else:
                for k in range(K):
                    for ctx, xd, gd in cs:
                        ctx.sweep_enqueue(xd, gd)
            for ctx, _, _ in cs:
                ctx.fetch_f()
            ms = 1e3 * (time.perf_counter() - t0) / K
            print(json.dumps({"contexts": parts, "batch_each": b, "host_stagger": stagger, "ms_per_sweep_of_the_whole_batch": round(ms, 3),
                              "sweeps_per_s": round(B / ms * 1e3, 1), "max_rel_dF_vs_one_context": float(np.max(np.abs(f - f_ref) / np.abs(f_ref)))}), flush=True)
        for ctx, _, _ in cs:
            ctx.close()


if __name__ == "__main__":
    main()
