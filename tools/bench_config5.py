#!/usr/bin/env python3
"""BASELINE configs[4]: Lorenz-96-sized recursions at D = 4096, RK4, S_t / Psi_t ROW-SHARDED over the GPUs of one node
with one all-to-all + one grouped all-gather (RCCL over xGMI) per RK stage, the whole step / stage loop and the
collectives inside libvgpa_hip.so (vgpa_shard_solve_fwd / _bwd, vgpa_amd/csrc/large_d.hip).

    python tools/bench_config5.py [--dim 4096] [--np 11] [--reps 2]                               # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_config5.py --gpus N [...]                                                        # N GPUs, strong scaling
    ... --sweep        # the FUSED sweep (vgpa_shard_sweep: F + gradient, time-parallel E_sde / gradient phases) instead of the
                       # two bare recursions

Every rank builds the same seeded inputs on its own GPU (A_t = 8 I + 0.05 N(0,1)/sqrt(D), symmetric dEsde_dS, sparse
jumps), runs the forward (m_t, S_t) and backward (lam_t, Psi_t) recursions over the same grid -- total work fixed, so
the per-N values give STRONG scaling -- and rank 0 prints one JSON line (time = max over ranks, barrier + synchronize
on both sides).  The full grid of the config (N = 10000 steps) does not fit any machine at D = 4096 (one (Np, D, D)
array = 1.34 TB); the step rate is grid-length independent, so a short grid measures it."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--np", dest="n_pts", type=int, default=11)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--method", default="rk4")
    ap.add_argument("--sweep", action="store_true", help="fused sweep (free energy + gradient) instead of the bare recursions")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from vgpa_amd import parallel as par
    from vgpa_amd.large_d import NativeShardedRecursion
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    par.init_from_env("nccl", local_rank)
    dev = torch.device("cuda", local_rank)
    d, n, dt = args.dim, args.n_pts, 0.01
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    f64 = dict(dtype=torch.float64, device=dev)
    a = torch.randn((n, d, d), generator=gen, **f64).mul_(0.05 / math.sqrt(d))
    a += 8.0 * torch.eye(d, **f64)
    b = 8.0 + torch.randn((n, d), generator=gen, **f64)
    m0 = 8.0 + torch.randn(d, generator=gen, **f64)
    s0, sigma = 0.2 * torch.eye(d, **f64), 4.0 * torch.eye(d, **f64)
    gs = torch.randn((n, d, d), generator=gen, **f64).mul_(1.0 / math.sqrt(d))
    gs = gs + gs.transpose(1, 2)
    gm = torch.randn((n, d), generator=gen, **f64)
    js, jm = torch.zeros((n, d, d), **f64), torch.zeros((n, d), **f64)
    for t in range(3, n, 4):
        js[t] = 0.5 * torch.eye(d, **f64)
        jm[t] = torch.randn(d, generator=gen, **f64)
    rec = NativeShardedRecursion(args.method, dt, d, n, rank=rank, world=world, device=local_rank)
    t_lo, t_hi = rec.time_slice
    if args.sweep:
        return fused_sweep(args, rec, a, b, m0, gen, rank, world, dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def once():
        mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
        lam, psi = rec.solve_bwd(a, gm, gs, jm, js)
        return mt, st, lam, psi

    mt, st, lam, psi = once()                                # warm-up
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        mt, st, lam, psi = once()
    barrier()
    elapsed = par.max_over_ranks((time.perf_counter() - t0) / args.reps, device="cuda")
    # every rank holds its own time slice: the last grid point of S lives on the last rank, Psi_0 on rank 0
    chk_t = torch.zeros(3, **f64)
    if t_lo <= n - 1 < t_hi:
        chk_t[0] = st[n - 1 - t_lo].abs().sum()
        chk_t[2] = (st[n - 1 - t_lo] - st[n - 1 - t_lo].T).abs().max()
    if t_lo == 0 and t_hi > 0:
        chk_t[1] = psi[0].abs().sum()
    if world > 1:
        dist.all_reduce(chk_t)
    chk = [float(v) for v in chk_t]
    if rank == 0:
        stages = {"euler": 1, "heun": 2, "rk2": 2, "rk4": 4}[args.method.lower()]
        flop = 2 * (n - 1) * stages * 2.0 * d ** 3          # fwd + bwd, one D^3 product per stage (symmetry)
        print(json.dumps({"config": f"BASELINE configs[4]: D={d} {args.method.upper()} recursions, row-sharded over {world} GPU(s), Np={n}",
                          "n_gpus": world, "D": d, "Np": n, "s_per_fwd_bwd": elapsed, "steps_per_s": 2 * (n - 1) / elapsed,
                          "tflops_aggregate": flop / elapsed / 1e12, "scaling": "strong",
                          "checks": {"sum|S_T|": chk[0], "sum|Psi_0|": chk[1], "asym(S_T)": chk[2]},
                          "collectives_per_stage": 0 if world == 1 else 2, "driver": "native (vgpa_shard_*)", "history": "time-sharded",
                          "all_gather_bytes_per_rank_per_stage": 0 if world == 1 else 8 * d * d // world}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def fused_sweep(args, rec, a, b, m0, gen, rank, world, dev):
    """F + gradient of one Lorenz-96 problem (theta = 8, Sigma = 4 I, S0 = 0.2 I, r = 1, H = I, an observation every 4th grid
    point) through vgpa_shard_sweep; strong scaling (the problem is fixed, the ranks split rows inside the recursions and grid
    points in the energy / gradient phases)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vgpa_amd import parallel as par
    d, n = args.dim, args.n_pts
    f64 = dict(dtype=torch.float64, device=dev)
    b = 8.0 * m0 + torch.randn((n, d), generator=gen, **f64)          # b_t = 8 m0 + noise keeps m_t near m0
    x = torch.cat((a.reshape(-1), b.reshape(-1)))
    del a
    obs_t = np.arange(3, n - 1, 4, dtype=np.int64)
    obs_y = (8.0 + torch.randn((max(obs_t.size, 1), d), generator=gen, **f64)).cpu().numpy()[:obs_t.size]
    sig, rdiag = np.full(d, 4.0), np.ones(d)
    s0 = 0.2 * np.eye(d)
    m0h = m0.cpu().numpy()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # as an optimisation calls it: fixed operands uploaded once, one pair of gradient arrays for every sweep (EXPERIMENTS.md s.14)
    problem = rec.prepare(8.0, sig, m0h, s0, obs_t, obs_y, rdiag, 0.0)
    f, ga, gb = rec.sweep(x, problem)                        # warm-up (allocates the sweep's buffers)

    def once():
        return rec.sweep(x, problem, out=(ga, gb))

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        f, ga, gb = once()
    barrier()
    elapsed = par.max_over_ranks((time.perf_counter() - t0) / args.reps, device="cuda")
    ga_t, gb_t = (torch.as_tensor(v, device=dev) for v in (ga, gb))      # (DeviceArray: __cuda_array_interface__, no copy)
    chk = torch.tensor([float(ga_t.abs().sum()), float(gb_t.abs().sum())], **f64)
    if world > 1:
        dist.all_reduce(chk)
    if rank == 0:
        stages = {"euler": 1, "heun": 2, "rk2": 2, "rk4": 4}[args.method.lower()]
        flop_rec = 2 * (n - 1) * stages * 2.0 * d ** 3
        print(json.dumps({"config": f"BASELINE configs[4]: D={d} {args.method.upper()} FUSED sweep (F + gradient), row-sharded recursions + "
                                    f"time-sharded energy / gradient over {world} GPU(s), Np={n}",
                          "n_gpus": world, "D": d, "Np": n, "s_per_sweep": elapsed, "sweeps_per_s": 1.0 / elapsed, "F": f,
                          "finite": bool(np.isfinite(f)), "recursion_tflop_per_sweep": flop_rec / 1e12, "scaling": "strong",
                          "checks": {"sum|gLa|": float(chk[0]), "sum|gLb|": float(chk[1])},
                          "collectives": "per RK stage: 1 all-to-all + 1 grouped all-gather; per sweep: 1 grouped all-gather of the "
                                         "E_sde terms + 1 of the observation jumps" if world > 1 else "none",
                          "driver": "native (vgpa_shard_sweep)", "gradient": "time-sharded"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
