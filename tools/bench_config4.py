#!/usr/bin/env python3
"""BASELINE configs[3]: Lorenz-96, D = 1024, RK4, N = 10000 (Np = 10001) on ONE MI355X -- the fused sweep
(free energy + gradient) in the time-chunked mode that keeps only x, S_t and the gradient resident (3 x 84 GB).

    python tools/bench_config4.py [D] [Np] [reps] [lib]      # `lib`: rocBLAS dgemm for the plain stage products (yardstick)

Inputs are generated on the device (torch is plumbing: memory + RNG): the build's own generator of SURVEY.md s.8d --
Sigma = 4 I, S0 = 0.2 I, m0 = 8 + N(0,1), A_t = 8 I + 0.05 N(0,1)/sqrt(D), b_t = 8 m0 + N(0,1), observation density 8
per time unit, r = 1, H = I.  If the three resident arrays do not fit into free HBM the grid is shortened and the Np
actually used is reported.  Prints one JSON line."""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import torch
    import vgpa_amd as va
    from vgpa_amd._lib import ExternalBuffer, FLAG_STREAM_LARGE_D, FLAG_LIBRARY_GEMM
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lib_gemm = len(sys.argv) > 4 and sys.argv[4] == "lib"
    dev = torch.device("cuda", 0)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    per_pt = 3 * (d * d + d) * 8                      # x, g, S (+ the vectors) per grid point
    n_fit = int((0.96 * free_b - 6e9) // per_pt)
    n = min(n_req, n_fit)
    dt = 0.01
    rng = np.random.default_rng(1)
    m0 = 8.0 + rng.standard_normal(d)
    obs_t = np.arange(12, n - 1, 12, dtype=np.int64)            # ~8 observations per time unit (dt = 0.01)
    obs_y = 8.0 + np.random.default_rng(4).standard_normal((obs_t.size, d))
    len_x = n * d * d + n * d
    # x and the gradient are torch tensors; the context consumes / fills them in place
    x = torch.empty(len_x, dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    eye8 = 8.0 * torch.eye(d, dtype=torch.float64, device=dev)
    step = 64
    for t0 in range(0, n, step):
        nc = min(step, n - t0)
        blk = torch.randn((nc, d, d), generator=gen, dtype=torch.float64, device=dev)
        blk.mul_(0.05 / math.sqrt(d)).add_(eye8)
        x[t0 * d * d:(t0 + nc) * d * d] = blk.reshape(-1)
    del blk
    bvec = torch.randn((n, d), generator=gen, dtype=torch.float64, device=dev)
    bvec.add_(8.0 * torch.as_tensor(m0, device=dev))
    x[n * d * d:] = bvec.reshape(-1)
    del bvec
    g = torch.empty(len_x, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx = va.Context("L96", "rk4", d, n, dt, sigma=4.0 * np.eye(d), theta=[8.0], m0=m0, s0=0.2 * np.eye(d), obs_t=obs_t,
                     obs_y=obs_y, obs_noise=np.eye(d), e0=0.0, flags=FLAG_STREAM_LARGE_D | (FLAG_LIBRARY_GEMM if lib_gemm else 0))
    assert ctx.streaming
    xb, gb = ExternalBuffer(x.data_ptr(), len_x), ExternalBuffer(g.data_ptr(), len_x)
    t0 = time.perf_counter()
    f = ctx.sweep_dev(xb, gb)                                   # warm-up (allocates the chunk buffers / workspaces)
    t_first = time.perf_counter() - t0
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        f = ctx.sweep_dev(xb, gb)
    t_sweep = (time.perf_counter() - t0) / reps
    pr = ctx.profile_end()
    # size-independent checks: finite, S_T symmetric, gradient finite, lam_0 finite
    s_last = ctx.fetch("st")[-1] if n <= 64 else None
    gsum, gmax = 0.0, 0.0
    for lo in range(0, len_x, 1 << 27):                          # chunked: no 80 GB temporaries
        part = g[lo:lo + (1 << 27)].abs()
        gsum += float(part.sum())
        gmax = max(gmax, float(part.max()))
    del part
    free_after, _ = torch.cuda.mem_get_info(dev)
    flop_rec = (n - 1) * 4 * 2.0 * d ** 3                       # one D^3 product per RK stage (symmetry), fwd or bwd
    flop_sweep = (n - 1) * 24.0 * d ** 3                        # 24 D^3 per grid point
    tag = "BASELINE configs[3]: " if (d == 1024 and n == 10001) else ""
    out = {"config": tag + "Lorenz96 D=%d RK4 Np=%d (requested %d), 1 x MI355X, time-chunked sweep" % (d, n, n_req),
           "D": d, "Np": n, "Np_requested": n_req, "F": f, "finite": bool(np.isfinite(f) and np.isfinite(gsum)),
           "grad_abs_sum": gsum, "grad_abs_max": gmax,
           "s_per_sweep": t_sweep, "sweeps_per_s": 1.0 / t_sweep, "first_sweep_s": t_first,
           "phase_ms": {"fwd": pr["fwd_ms"] / reps, "obs": pr["energy_ms"] / reps,
                        "chunked energy+bwd+grad": pr["bwd_ms"] / reps, "reduce": pr["grad_ms"] / reps},
           "fwd_tflops": flop_rec / (pr["fwd_ms"] / reps) / 1e9,
           "sweep_tflops": flop_sweep / t_sweep / 1e12,
           "resident_GB": 3 * (d * d + d) * n * 8 / 1e9, "hbm_free_before_GB": free_b / 1e9, "hbm_free_after_GB": free_after / 1e9,
           "stage_gemm": "rocBLAS dgemm (VGPA_FLAG_LIBRARY_GEMM)" if lib_gemm else "hand-written k_gemm"}
    if s_last is not None:
        out["S_T_asym"] = float(np.abs(s_last - s_last.T).max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
