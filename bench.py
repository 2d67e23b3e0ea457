#!/usr/bin/env python3
"""
bench.py -- fwd+bwd sweeps/sec of the VGPA variational smoothing sweep on MI355X.

One "step" = one pass of the hot path (vgpa_sweep: forward moment ODE -> E_obs / E_sde terms -> backward
Lagrange ODE -> gradient w.r.t. (A_t, b_t)) over one BATCH of `--batch` independent problems that share the
BASELINE configuration 3 (Lorenz-96, D=40, RK4, Np=1001, seed 31415926535) and differ in their variational
parameters x.  Inputs (x) and outputs (gradient) stay resident in HBM; F comes back to the host once per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, independent problems per rank (the path shards over problems; no data-path
collective), barrier + device sync on both sides of the timed region, MAX over ranks, weak scaling.  Started WITHOUT a
launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the script starts its N ranks itself -- torch.distributed.run as a CHILD
process, before this process has made any GPU call -- and exits with the child's code; a WORLD_SIZE that disagrees with --gpus is
an error, never a silent one-GPU run.

Secondary block `config4` (rank 0; --no-config4 skips it): BASELINE configs[3]'s matrix size -- ONE Lorenz-96 problem, D = 1024, RK4, the
resident fused sweep on a 33-point grid, checked against the committed oracle anchors of that size.
Secondary block `config5` (every N; --no-config5 skips it): BASELINE configs[4]'s matrix size -- ONE Lorenz-96 problem, D = 4096,
RK4, short grid -- through vgpa_shard_sweep_sharded: S_t / Psi_t row-sharded over the N ranks with RCCL collectives per RK stage,
energy / gradient phases time-parallel, x and the gradient memory-sharded.  The problem is fixed, so the per-N values of
`config5.s_per_sweep` give STRONG scaling; the headline `value` stays the D = 40 metric.  At N > 1 the block runs in CHILD processes
of the ranks (own process group, started before the rank touches the GPU, reaped before the headline measurement): it is the only
part of the bench that needs several real RCCL ranks, so a crash or hang there must not cost the headline line (config5_children).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import os
import sys
import json
import time
import argparse

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 FMA/clk x 2 x 2.4 GHz; measured 64 clk per v_mfma_f64_16x16x4 (profiles/)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="independent problems per GPU per step")
    ap.add_argument("--np", dest="n_pts", type=int, default=1001)
    ap.add_argument("--dim", type=int, default=40)
    ap.add_argument("--method", default="RK4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-problem", action="store_true",
                    help="skip the single-problem latency probe (so that a kernel profile of the run holds batched launches only)")
    ap.add_argument("--generic", action="store_true", help="force the generic (non-MFMA) stepping kernels")
    ap.add_argument("--keep-psi", action="store_true", help="comparison runs: VGPA_FLAG_KEEP_PSI (the backward kernel stores Psi_t, "
                    "the gradient assembly re-reads A_t) instead of the default Q''_t stream")
    ap.add_argument("--no-config5", action="store_true", help="skip the secondary D = 4096 row-sharded block")
    ap.add_argument("--config5-np", type=int, default=64, help="grid points of the D = 4096 block: a multiple of 8, so that the "
                    "time-parallel energy / gradient phases are balanced at N = 1, 2, 4, 8 (vgpa_shard_time_slice); 64 = eight per rank "
                    "at N = 8, so that the 64-step chain of the blocked Cholesky's diagonal kernel is amortised over a batch")
    ap.add_argument("--config5-dim", type=int, default=4096)
    ap.add_argument("--config5-timeout", type=float, default=300.0, help="N > 1: seconds the config-5 child processes may take")
    ap.add_argument("--config5-child", action="store_true", help=argparse.SUPPRESS)     # internal: see config5_children
    ap.add_argument("--no-config4", action="store_true", help="skip the secondary D = 1024 block (BASELINE configs[3]'s matrix size on a short grid)")
    ap.add_argument("--no-config2", action="store_true", help="skip the secondary Lorenz-63 block (BASELINE configs[1], the HBM-bound small-D path)")
    ap.add_argument("--config2-batch", type=int, default=65536, help="independent Lorenz-63 problems of the config-2 block: 1024 waves of 64 -- "
                    "one per SIMD, what the lane kernels' registers admit (any multiple fills the chip evenly)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a child process group (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1) and hand its exit code back.  Nothing in THIS process has touched the GPU (no torch
    import, no HIP call): a process that holds the GPU is never replaced by another program."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] no launcher (WORLD_SIZE unset): starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def _cpu_sweep(job):
    """One oracle sweep in a worker process (BLAS pinned to one thread): the all-core CPU baseline runs one per core."""
    prob, x, faithful = job
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    from oracle import vgpa_oracle as vo
    t0 = time.perf_counter()
    f, _, _ = vo.sweep(prob, x, faithful=faithful)
    return time.perf_counter() - t0, float(f)


def cpu_baseline(p, x, method, d, n_pts, dt):
    """The numpy oracle on the host cores, BEFORE this process touches the GPU (the pool forks).  `value`: one sweep on one
    core in the reference's operation order (faithful mode); `many_cores`: one such sweep per core of this job's share of the host
    (`cores_used` workers; `host_cores` says what the machine has), concurrently --
    independent problems are how the path parallelises on a CPU (a 40 x 40 sweep has nothing for BLAS threads to do)."""
    import multiprocessing as mp
    from oracle import vgpa_oracle as vo
    z = dict(model="L96", method=method, dt=dt, theta=8.0, sigma=p["model"].sigma, m0=p["m0"], s0=p["s0"],
             mu0=p["mu0"], tau0=p["tau0"], obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"],
             time_window=p["model"].time_window)
    prob = vo.Problem.from_fixture({k: np.asarray(val) for k, val in z.items()})
    t_faith, f_cpu = _cpu_sweep((prob, x, True))
    t_lean, _ = _cpu_sweep((prob, x, False))
    print(f"[bench] cpu baseline: one core {t_faith:.2f} s (faithful), {t_lean:.2f} s (lean)", file=sys.stderr, flush=True)
    # (a GPU box gives one GPU's job a share of the host: 16 cores; more workers than that only time-slice)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("VGPA_BENCH_CPU_WORKERS", "16"))))
    all_cores = None
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            tw = time.perf_counter()
            res = pool.map_async(_cpu_sweep, [(prob, x, True)] * cores).get(timeout=20.0 * t_faith + 60.0)
            wall = time.perf_counter() - tw
        all_cores = {"cores_used": cores, "value": cores / wall, "unit": "sweeps/s", "seconds": wall,
                     "how": f"{cores} concurrent single-threaded oracle processes, one faithful sweep each",
                     "slowest_worker_s": max(r[0] for r in res)}
    except Exception as exc:                                  # (a box that forbids fork: keep the one-core figure)
        all_cores = {"cores_used": cores, "value": None, "error": repr(exc)}
    return {"value": 1.0 / t_faith, "unit": "sweeps/s", "cores": 1, "kind": "port",
            "sample": f"1 full sweep of the same workload (L96 D={d} Np={n_pts}), numpy oracle in faithful mode (same "
                      f"per-step operations as the reference), one core; many_cores: one such sweep on each of cores_used cores",
            "seconds": t_faith, "lean_value": 1.0 / t_lean, "lean_seconds": t_lean, "host_cores": os.cpu_count(),
            "many_cores": all_cores, "F_cpu": f_cpu}


def peak_context():
    path = os.path.join(ROOT, "profiles", "peak_context.json")
    try:
        ctx = json.load(open(path))
    except (OSError, ValueError):
        return {"fp64_matrix_nominal_TFLOPs": FP64_PEAK_TFLOPS, "error": "profiles/peak_context.json is missing (tools/peak_context.py)"}
    ctx["measured_in_this_run"] = False
    ctx["file"] = "profiles/peak_context.json"
    return ctx


def config2_block(args, local_rank):
    """BASELINE configs[1]: Lorenz-63, D = 3, RK4, Np = 1001, the reference's seeded inputs -- the small-D path, which IS bound by
    HBM.  A batch of independent problems on the lane-per-problem kernels: forward moments (k_fwd_lane), observation
    terms (k_obs_lane), ONE fused backward pass (k_sweep_lane: closed-form E_sde terms in registers, (lam, Psi) recursion, gradient, F);
    (m_t, S_t) live in a time-major array of the context (problem fastest: coalesced straight into registers), x and the gradient
    pass through LDS in chunks of six grid points.
    Roofline on SURVEY 8(d)'s algorithmic bytes 8 Np (5 D^2 + 6 D) per sweep against 8 TB/s; problem 0 must reproduce the
    reference's anchor; the numpy oracle is timed beside it (one sweep, one core).  Never raises."""
    try:
        import vgpa_amd as va
        from helpers import build_problem
        from oracle import vgpa_oracle as vo
        d, n_pts, dt, B = 3, 1001, 0.01, args.config2_batch
        p = build_problem("L63", "RK4", (n_pts - 1) * dt, dt, None)
        x0 = p["vgp"].initialization()
        len_x = x0.size
        e0 = float(p["kl0"](p["m0"], p["s0"]))
        x_first = x0 + 0.05 * np.random.default_rng(0).standard_normal(len_x)          # problem 0: the reference's perturbed anchor input
        # CPU oracle beside it (the same sweep, one core)
        z = dict(model="L63", method="RK4", dt=dt, theta=np.asarray(p["model"].theta), sigma=p["model"].sigma, m0=p["m0"], s0=p["s0"],
                 mu0=p["mu0"], tau0=p["tau0"], obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], time_window=p["model"].time_window)
        prob = vo.Problem.from_fixture({k: np.asarray(val) for k, val in z.items()})
        t0 = time.perf_counter()
        f_cpu, _, _ = vo.sweep(prob, x_first, faithful=False)
        t_cpu = time.perf_counter() - t0
        ctx = va.Context("L63", "RK4", d, n_pts, dt, sigma=p["model"].sigma, theta=p["model"].theta, m0=p["m0"], s0=p["s0"],
                         obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=B, device=local_rank)
        x_dev, g_dev = ctx.alloc(B * len_x), ctx.alloc(B * len_x)
        noise = 0.05 * np.random.default_rng(1).standard_normal((61, len_x))           # 61 further inputs, repeated over the batch
        rows = np.vstack([x_first[None, :], x0[None, :] + noise])
        piece = 62 * 64
        tile = np.tile(rows, (64, 1))
        for i0 in range(0, B, piece):
            k = min(piece, B - i0)
            x_dev.upload_at(i0 * len_x, tile[:k])
        for _ in range(2):
            ctx.sweep_enqueue(x_dev, g_dev)
            ctx.fetch_f()
        steps = 10
        t0 = time.perf_counter()           # the rate: ten steps without the phase events ...
        for _ in range(steps):
            ctx.sweep_enqueue(x_dev, g_dev)
            f = ctx.fetch_f()
        secs = (time.perf_counter() - t0) / steps
        ctx.profile_begin()                # ... the per-kernel times: ten more with them (five event records per step cost up to 0.3 ms of a 7 ms step)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.sweep_enqueue(x_dev, g_dev)
            f = ctx.fetch_f()
        secs_events = (time.perf_counter() - t0) / steps
        pr = ctx.profile_end()
        anchors = json.load(open(os.path.join(ROOT, "tests", "golden", "anchors.json")))["l63_rk4_full_p"]
        g0 = g_dev.download_at(0, len_x)
        err_f = abs(f[0] - anchors["F"]) / abs(anchors["F"])
        err_g = abs(float(np.linalg.norm(g0)) - anchors["grad_norm"]) / anchors["grad_norm"]
        ctx.close()
        alg = 8.0 * n_pts * (5 * d * d + 6 * d)
        fwd_s, bwd_s = pr["fwd_ms"] / steps * 1e-3, pr["bwd_ms"] / steps * 1e-3
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
        gbs = alg * B / secs / 1e9
        return {"workload": f"Lorenz63 D=3, RK4, Np={n_pts} (BASELINE configs[1]), {B} independent problems, one lane per problem",
                "sweeps_per_s": B / secs, "ms_per_step": 1e3 * secs, "ms_per_step_with_phase_events": 1e3 * secs_events, "batch": B,
                "phase_ms_per_step": {"fwd (k_fwd_lane)": 1e3 * fwd_s, "obs (k_obs_lane)": pr["energy_ms"] / steps,
                                      "fused E_sde + bwd + grad + F (k_sweep_lane)": 1e3 * bwd_s},
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "alg_bytes_per_sweep": alg, "alg_bytes_per_step": alg * B,
                             "traffic": tj.get(f"L63_sweep_B{B}"),
                             "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_small.sh; not measured in this run)",
                             "kernels": {"k_fwd_lane": {"launch_ms": 1e3 * fwd_s, "alg_GBs": 8.0 * n_pts * (2 * d * d + 2 * d) * B / fwd_s / 1e9},
                                         "k_sweep_lane": {"launch_ms": 1e3 * bwd_s, "alg_GBs": 8.0 * n_pts * (3 * d * d + 4 * d) * B / bwd_s / 1e9}}},
                "cpu_baseline": {"value": 1.0 / t_cpu, "unit": "sweeps/s", "cores": 1, "kind": "port", "seconds": t_cpu,
                                 "sample": "one full sweep of problem 0, numpy oracle", "gpu_vs_cpu_F_rel_err": abs(f[0] - f_cpu) / abs(f_cpu)},
                "parity_check_rel_err_F": err_f, "parity_check_rel_err_grad_norm": err_g,
                "parity_ok": bool(max(err_f, err_g) < 1e-9)}
    except Exception as exc:                                 # noqa: BLE001 - the headline line must still be printed
        return {"error": repr(exc)}


def config4_block(args, local_rank):
    """BASELINE configs[3]'s matrix size: ONE Lorenz-96 problem, D = 1024, RK4, on a grid of 33 points (the full grid, Np = 10 001,
    is 252 GB resident and 6 s per sweep: tools/bench_config4.py, profiles/*config4_D1024_Np10001.json) -- the resident fused sweep
    through the per-stage kernels of large_d.hip (fp64-MFMA stage products + the element-wise stage), inputs resident in HBM.
    The inputs are the seeded ones of tests/golden/anchors_d1024_np33.json (tools/gen_d4096_anchor.inputs_grid): F and per-grid-
    point gradient norms must reproduce those oracle anchors.  Roofline: fp64 matrix pipe, nominal 2 x 4 x 2 D^3 flop per step of
    the two recursions.  Never raises."""
    try:
        import vgpa_amd as va
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from gen_d4096_anchor import inputs_grid
        a = json.load(open(os.path.join(ROOT, "tests", "golden", "anchors_d1024_np33.json")))
        d, n = a["D"], a["Np"]
        x, m0, s0, sig, obs_t, obs_y, rdiag = inputs_grid(d, n)
        ctx = va.Context("L96", a["method"], d, n, a["dt"], sigma=np.diag(sig), theta=[8.0], m0=m0, s0=s0, obs_t=obs_t, obs_y=obs_y,
                         obs_noise=np.diag(rdiag), e0=0.0, device=local_rank)
        x_dev, g_dev = ctx.alloc(x.size), ctx.alloc(x.size)
        x_dev.upload(x)
        ctx.sweep_dev(x_dev, g_dev)
        reps = 4
        ctx.profile_begin()
        t0 = time.perf_counter()
        for _ in range(reps):
            f = ctx.sweep_dev(x_dev, g_dev)
        secs = (time.perf_counter() - t0) / reps
        pr = ctx.profile_end()
        f = float(np.atleast_1d(f)[0])
        g = g_dev.download_at(0, x.size)
        ga = g[:n * d * d].reshape(n, d, d)
        err_f = abs(f - a["F_minus_E0"]) / abs(a["F_minus_E0"])
        err_g = max(abs(float(np.linalg.norm(ga[t])) - a["grad_a_fro"][t]) / a["grad_a_fro"][t] for t in range(n))
        ctx.close()
        flop_rec = (n - 1) * 4 * 2.0 * d ** 3
        fwd_s, bwd_s = pr["fwd_ms"] / reps * 1e-3, pr["bwd_ms"] / reps * 1e-3
        return {"workload": f"Lorenz96 D={d}, RK4, Np={n}: ONE problem, resident fused sweep (BASELINE configs[3]'s matrix size; the full "
                            "grid: tools/bench_config4.py)",
                "ms_per_sweep": 1e3 * secs, "sweeps_per_s": 1.0 / secs,
                "phase_ms": {"fwd": 1e3 * fwd_s, "energy+obs": pr["energy_ms"] / reps, "bwd": 1e3 * bwd_s, "grad": pr["grad_ms"] / reps},
                "roofline": {"bound": "mfma", "kernel": "forward recursion (k_gemm_v + k_stage_sym per RK stage)", "achieved": flop_rec / fwd_s / 1e12,
                             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop_rec / fwd_s / 1e12 / FP64_PEAK_TFLOPS,
                             "bwd_achieved": flop_rec / bwd_s / 1e12,
                             "note": "nominal 2 D^3 flop per stage product; kernel stats and matrix-pipe counters: profiles/r05d_stage_two-kernel_D1024_*.csv"},
                "parity_check_rel_err_F": err_f, "parity_check_max_rel_err_grad_norm_per_grid_point": err_g,
                "parity_ok": bool(max(err_f, err_g) < 1e-9)}
    except Exception as exc:                                 # noqa: BLE001 - the headline line must still be printed
        return {"error": repr(exc)}


def config5_block(args, rank, world, local_rank, rehearse):
    """One Lorenz-96 problem at BASELINE configs[4]'s matrix size on ALL ranks of the job: the fused sweep (free energy + gradient)
    with S_t / Psi_t row-sharded (vgpa_shard_sweep_sharded: RCCL all-to-all + pipelined gather per RK stage), x and the gradient
    memory-sharded.  Strong scaling: the problem does not change with N.  Never raises: a failure is reported in the block."""
    import math
    import torch
    import torch.distributed as dist
    from vgpa_amd import parallel as par
    d, n = args.config5_dim, args.config5_np
    if d % world:
        return {"skipped": f"D={d} is not a multiple of {world} ranks"}
    try:
        from vgpa_amd.large_d import NativeShardedRecursion
        from vgpa_amd._lib import SHARD_OPT_TIMEOUT_MS
        dev = torch.device("cuda", local_rank)
        f64 = dict(dtype=torch.float64, device=dev)
        comm = None
        if rehearse and world > 1:              # several ranks on ONE GPU: no RCCL communicator possible; gloo, staged through the host
            from vgpa_amd.large_d import HostStagedComm
            comm = HostStagedComm().table
        rec = NativeShardedRecursion("rk4", 0.01, d, n, rank=rank, world=world, device=local_rank, comm=comm)
        rec.set_option(SHARD_OPT_TIMEOUT_MS, 120000)
        lo, hi = rec.time_slice
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)                                   # every rank draws the same problem and keeps its own grid points
        m0 = 8.0 + torch.randn(d, generator=gen, **f64)
        a_own = torch.empty((max(hi - lo, 1), d, d), **f64)
        for t in range(n):
            a_t = torch.randn((d, d), generator=gen, **f64).mul_(0.05 / math.sqrt(d))
            a_t.diagonal().add_(8.0)
            if lo <= t < hi:
                a_own[t - lo] = a_t
        del a_t
        b_all = 8.0 * m0 + torch.randn((n, d), generator=gen, **f64)
        obs_t = np.arange(3, n - 1, 4, dtype=np.int64)
        obs_y = (8.0 + torch.randn((max(obs_t.size, 1), d), generator=gen, **f64)).cpu().numpy()[:obs_t.size]
        sig, rdiag, s0, m0h = np.full(d, 4.0), np.ones(d), 0.2 * np.eye(d), m0.cpu().numpy()

        def sync():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # as an optimisation calls it: the problem's fixed operands uploaded once, ONE pair of gradient arrays written by every sweep
        # (a fresh 8.6 GB gLa_own per sweep, the previous one freed, was host time inside the timed sweeps until the end of round 5)
        problem = rec.prepare(8.0, sig, m0h, s0, obs_t, obs_y, rdiag, 0.0)
        f, ga, gb = rec.sweep_sharded(a_own[:hi - lo], b_all[lo:hi], problem)      # warm-up: allocates the sweep's buffers

        def once():
            return rec.sweep_sharded(a_own[:hi - lo], b_all[lo:hi], problem, out=(ga, gb))

        reps = 5
        sync()
        t0 = time.perf_counter()
        each = []
        for _ in range(reps):
            t1 = time.perf_counter()
            f, ga, gb = once()
            each.append(time.perf_counter() - t1)           # (a sweep ends with the ranks' agreement: the host has waited for it)
        sync()
        secs_mean = par.max_over_ranks((time.perf_counter() - t0) / reps, device="cpu" if rehearse else "cuda")
        # the MEDIAN of the five sweeps (each the max over ranks; three until the last run of round 5, where TWO of three were long: 1.68, 1.91, 1.44 s): in five of some twenty runs of round 5 one sweep of this block took
        # 0.3-0.5 s longer on the host's clock than its GPU phases add up to (never reproduced on purpose, EXPERIMENTS.md s.14); all
        # three times and the mean are printed beside it
        each_max = [par.max_over_ranks(v, device="cpu" if rehearse else "cuda") for v in each]
        secs = sorted(each_max)[len(each_max) // 2]
        # (the driver returns library-owned device arrays; torch wraps them without a copy through __cuda_array_interface__)
        ga_t, gb_t = torch.as_tensor(ga, device=dev), torch.as_tensor(gb, device=dev)
        chk = torch.tensor([float(ga_t.abs().sum()), float(gb_t.abs().sum())], dtype=torch.float64, device="cpu" if rehearse else dev)
        del ga_t, gb_t
        if world > 1:
            dist.all_reduce(chk)
        # where the sweep's time goes: the phases of the last timed sweep (HIP events on the shard's stream; MAX over ranks, and
        # rank 0's own), the two per-stage collectives alone, and ONE recursion stage with and without its collectives -- the
        # difference is the communication the pipelined schedule does not hide
        ph = rec.phase_ms()
        names = list(ph)
        pht = torch.tensor([ph[k] for k in names], dtype=torch.float64, device="cpu" if rehearse else dev)
        npts = torch.zeros(world, dtype=torch.float64, device="cpu" if rehearse else dev)
        npts[rank] = hi - lo
        if world > 1:
            dist.all_reduce(pht, op=dist.ReduceOp.MAX)
            dist.all_reduce(npts)
        a2a_ms, gather_ms = rec.time_collectives(10)
        stage_with = rec.time_stage(10, True)
        stage_without = rec.time_stage(10, False)
        if world > 1:
            st_t = torch.tensor([stage_with, stage_without], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(st_t, op=dist.ReduceOp.MAX)
            stage_with, stage_without = float(st_t[0]), float(st_t[1])
        chunks = rec.gather_chunks
        rccl_ranks = rec.rccl_ranks                           # ncclCommCount of the communicator the sweep ran on
        rccl_comms = rec.rccl_communicators                   # 2: compute-stream collectives and the gather's point-to-point groups apart
        rec.close()
        flop = 24.0 * d ** 3 * n                             # nominal flop of one fused sweep
        return {"workload": f"Lorenz96 D={d}, RK4, Np={n}: ONE problem, fused sweep (free energy + gradient), S_t / Psi_t row-sharded, "
                            f"energy / gradient time-parallel, x and gradient memory-sharded (BASELINE configs[4] matrix size; its "
                            f"full grid fits no node)",
                "n_gpus": world, "rccl_ranks": rccl_ranks, "rccl_communicators": rccl_comms, "rccl_ranks_how": "ncclCommCount of the shard's communicator (0: one rank or host-staged rehearsal)",
                "grid_points_per_rank": [int(v) for v in npts.tolist()],
                "transport": "host-staged gloo (REHEARSAL: not a measurement)" if comm is not None else ("RCCL" if world > 1 else "none"),
                "scaling": "strong", "s_per_sweep": secs, "s_per_sweep_how": "median of five timed sweeps, each the max over ranks", "s_per_sweep_mean": secs_mean,
                "s_each_sweep_rank0": each,
                "recursion_steps_per_s": 2 * (n - 1) / secs, "aggregate_tflops_nominal": flop / secs / 1e12,
                "frac_of_fp64_peak_per_gpu": flop / secs / 1e12 / world / FP64_PEAK_TFLOPS,
                "schedule": f"pipelined gather, {chunks} sub-blocks, second stream" if chunks else
                            ("serial (one grouped all-gather per stage)" if world > 1 else "one rank: no collective"),
                "phase_ms_max_over_ranks": {k: float(v) for k, v in zip(names, pht.tolist())},
                "phase_ms_rank0": ph,
                # the time-parallel phases per grid point of the busiest rank: energy_obs does not shrink below the 64-step diagonal-block
                # chain of the blocked Cholesky / inverse (69 us x 64 per batch at D = 4096) however few grid points a rank owns
                "ms_per_own_grid_point": {k: float(v) / max(float(npts.max()), 1.0) for k, v in zip(names, pht.tolist()) if k in ("energy_obs", "gradient")},
                "phase_how": "vgpa_shard_phase_ms: HIP events on the shard's stream around the phases of the last timed sweep; the recursions "
                             "(row-sharded, one collective pair per RK stage) scale with the stage time, energy_obs and gradient (time-parallel) with "
                             "grid_points_per_rank",
                "per_stage_ms": {"stage_with_collectives": stage_with, "stage_compute_only": stage_without,
                                 "collective_exposed": stage_with - stage_without,
                                 "collective_hidden": max(0.0, a2a_ms + gather_ms - (stage_with - stage_without)),
                                 "how": "vgpa_shard_time_stage: one forward RK4 stage as the driver issues it (K-chunk products waiting for the "
                                        "previous gather's sub-blocks, all-to-all, stage kernel, gather), 10 back to back, with and without the "
                                        "collectives; MAX over ranks"},
                "per_stage_collective_ms": {"all_to_all": a2a_ms, "gather": gather_ms,
                                            "how": "vgpa_shard_time_collectives: the stage's two collectives alone, same buffers and streams"},
                "F": f, "finite": bool(np.isfinite(f)), "checks": {"sum|gLa|": float(chk[0]), "sum|gLb|": float(chk[1])}}
    except Exception as exc:                                 # noqa: BLE001 - the headline line must still be printed
        return {"error": repr(exc)}


def config5_children(args, rank):
    """N > 1: the config-5 block puts more than one rank through RCCL -- a path no one-GPU box can exercise -- so it runs in CHILD
    processes of the ranks (one per rank, a process group of their own on a neighbouring port), started BEFORE the rank has touched
    the GPU and reaped before the headline measurement begins: a crash, a hang (bounded by --config5-timeout; the rank kills the
    exact child it started) or a communicator error there ends up as {"error": ...} in the block and cannot take the headline
    line with it.  Returns the block (rank 0) or None."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29500"))
    env = dict(os.environ)
    env["MASTER_PORT"] = str(port + 17 if port + 17 < 65000 else port - 17)
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:      # (with TORCHELASTIC_USE_AGENT_STORE the children would look for the
        del env[k]                                                   #  launcher's store on their port instead of opening their own)
    cmd = [sys.executable, os.path.abspath(__file__), *[a for a in sys.argv[1:] if a != "--config5-child"], "--config5-child"]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
    try:
        out, _ = proc.communicate(timeout=args.config5_timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": f"config-5 child processes did not finish within {args.config5_timeout:.0f} s (killed)"} if rank == 0 else None
    if rank != 0:
        return None
    for line in reversed((out or "").strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": f"config-5 child of rank 0 ended with code {proc.returncode} and no result"}


def config5_child_main(args):
    """Body of one config-5 child process (see config5_children): its own process group, the block, rank 0 prints it."""
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"])
    if os.environ.get("VGPA_BENCH_LAUNCH_ONLY") == "1":       # CPU test of the child mechanism: rendezvous on the children's port, report, leave
        if os.environ.get("VGPA_BENCH_C5_TEST_HANG") == "1":
            time.sleep(3600)
        import torch.distributed as dist
        from vgpa_amd import parallel as par
        par.init_from_env("gloo")
        par.barrier()
        seen = par.max_over_ranks(float(rank)) + 1.0
        if rank == 0:
            print(json.dumps({"child": "ok", "ranks_seen": int(seen), "port": os.environ["MASTER_PORT"]}), flush=True)
        dist.destroy_process_group()
        return
    import torch
    import torch.distributed as dist
    rehearse = os.environ.get("VGPA_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from vgpa_amd import parallel as par
    par.init_from_env("gloo" if rehearse else "nccl", None if rehearse else local_rank)
    c5 = config5_block(args, rank, world, local_rank, rehearse)
    if rank == 0:
        c5["process"] = "child processes of the ranks, own process group (bench.py::config5_children)"
        print(json.dumps(c5), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (n_gpus in the JSON line is the number of "
                         f"ranks that ran)")
    if args.config5_child:
        return config5_child_main(args)
    if os.environ.get("VGPA_BENCH_LAUNCH_ONLY") == "1":       # CPU test of the launcher: rendezvous, report, leave (no GPU work)
        import torch.distributed as dist
        from vgpa_amd import parallel as par
        c5 = config5_children(args, rank) if (world > 1 and not args.no_config5) else None
        par.init_from_env("gloo")
        par.barrier()
        seen = par.max_over_ranks(float(rank)) + 1.0
        if rank == 0:
            print(json.dumps({"launcher": "ok", "n_gpus": world, "ranks_seen": int(seen), "config5": c5}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # N > 1: the secondary D = 4096 block first, in child processes (nothing here has touched the GPU yet)
    c5_early = None
    c5_in_children = world > 1 and not args.no_config5 and not args.generic
    if c5_in_children:
        c5_early = config5_children(args, rank)

    # host-side inputs first (numpy only), and with them the CPU baseline: its worker pool forks, which must happen
    # before this process initialises the GPU
    from helpers import build_problem
    d, n_pts, dt = args.dim, args.n_pts, 0.01
    tf = (n_pts - 1) * dt
    p = build_problem("L96", args.method, tf, dt, d)          # seed 31415926535: the reference's config-3 inputs
    x0 = p["vgp"].initialization()
    x_first = x0 + 0.05 * np.random.default_rng(0).standard_normal(x0.size)     # global problem 0
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(p, x_first, args.method, d, n_pts, dt)
        print(f"[bench] cpu baseline: many cores {cpu['many_cores']}", file=sys.stderr, flush=True)

    # torch is plumbing only (process group, barrier, device sync).  It must be imported BEFORE libvgpa_hip.so
    # is loaded so that both share one HIP runtime (same SONAME, see vgpa_amd/_lib.py).
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (vgpa_amd has no CPU fallback)")
    # Rehearsal switch (not used by the driver): VGPA_BENCH_REHEARSE=1 runs all ranks on GPU 0 with the gloo backend so
    # that the N > 1 control flow can be exercised on a one-GPU box.
    rehearse = os.environ.get("VGPA_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from vgpa_amd import parallel as par
    par.init_from_env("gloo" if rehearse else "nccl", None if rehearse else local_rank)   # RCCL; no-op at WORLD_SIZE == 1

    import vgpa_amd as va
    from vgpa_amd._lib import FLAG_FORCE_GENERIC, FLAG_KEEP_PSI

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    v = p["vgp"]
    assert v.dim_n == n_pts, (v.dim_n, n_pts)
    len_x = x0.size
    B = args.batch
    flags = (FLAG_FORCE_GENERIC if args.generic else 0) | (FLAG_KEEP_PSI if args.keep_psi else 0)
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    ctx = va.Context("L96", args.method, d, n_pts, dt, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"],
                     obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=B,
                     device=local_rank, flags=flags)
    lo, hi = par.shard_range(B * world, rank, world)        # this rank's slice of the global problem list
    assert hi - lo == B
    xb = np.empty((B, len_x))
    for i in range(B):                                       # global problem 0 is exactly x0 + 0.05 N(0,1), rng(0)
        gi = lo + i
        r = np.random.default_rng(0) if gi == 0 else np.random.default_rng(1000 + gi)
        xb[i] = x0 + 0.05 * r.standard_normal(len_x)
    x_dev, g_dev = ctx.alloc(B * len_x), ctx.alloc(B * len_x)
    x_dev.upload(xb)

    for _ in range(args.warmup):
        ctx.sweep_enqueue(x_dev, g_dev)
        ctx.fetch_f()
    barrier()
    ctx.profile_begin()
    t0 = time.perf_counter()
    f_last = None
    for _ in range(args.steps):
        ctx.sweep_enqueue(x_dev, g_dev)
        f_last = ctx.fetch_f()                               # F of every problem returns to the host each step
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_end()

    elapsed = par.max_over_ranks(elapsed, device="cpu" if rehearse else "cuda")

    # ---- correctness guard inside the bench: problem 0 of rank 0 must reproduce the reference's anchors -- the free energy AND
    # the norm of the gradient the timed launches left in g_dev (per-problem dot products on the device)
    check = check_g = None
    if rank == 0 and n_pts == 1001 and d == 40 and args.method.upper() == "RK4":
        anchors = json.load(open(os.path.join(ROOT, "tests", "golden", "anchors.json")))
        f_ref, g_ref = anchors["l96d40_rk4_full_p"]["F"], anchors["l96d40_rk4_full_p"]["grad_norm"]
        check = abs(np.atleast_1d(f_last)[0] - f_ref) / abs(f_ref)
        check_g = abs(float(np.sqrt(np.atleast_1d(ctx.vdot(g_dev, g_dev))[0])) - g_ref) / g_ref
        if max(check, check_g) > 1e-9 and os.environ.get("VGPA_BENCH_NO_CHECK") != "1":   # (diagnostic builds only)
            raise SystemExit(f"bench result is WRONG: F={np.atleast_1d(f_last)[0]!r} vs reference {f_ref!r}; "
                             f"|grad| rel. err {check_g:.3e}")

    # ---- single-problem latency (what one SCG evaluation costs), rank 0, outside the timed region
    single = None
    if rank == 0 and not args.no_single_problem:
        c1 = va.Context("L96", args.method, d, n_pts, dt, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"],
                        obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=1,
                        device=local_rank, flags=flags)
        x1, g1 = c1.alloc(len_x), c1.alloc(len_x)
        x1.upload(xb[0])
        for _ in range(2):
            c1.sweep_dev(x1, g1)
        c1.profile_begin()
        ts = time.perf_counter()
        reps = 10
        for _ in range(reps):
            c1.sweep_dev(x1, g1)
        t_single = (time.perf_counter() - ts) / reps
        pr1 = c1.profile_end()
        single = {"ms_per_sweep": 1e3 * t_single, "sweeps_per_s": 1.0 / t_single,
                  "fwd_ms": pr1["fwd_ms"] / reps, "energy_ms": pr1["energy_ms"] / reps,
                  "bwd_ms": pr1["bwd_ms"] / reps, "grad_ms": pr1["grad_ms"] / reps}
        c1.close()

    # ---- secondary block: BASELINE configs[1] (Lorenz-63), the HBM-bound small-D path; rank 0 only, outside the timed region
    c2 = None
    if rank == 0 and not args.no_config2 and not args.generic:
        ctx.close()
        del x_dev, g_dev
        ctx = None
        c2 = config2_block(args, local_rank)
    c4 = None
    if rank == 0 and not args.no_config4 and not args.generic:
        if ctx is not None:
            ctx.close()
            del x_dev, g_dev
            ctx = None
        c4 = config4_block(args, local_rank)

    # ---- secondary block: BASELINE configs[4]'s matrix size through the row-sharded driver (every rank takes part)
    # (one rank: in this process, behind the headline measurement; N > 1: already measured, in child processes -- see above)
    c5 = c5_early
    if not args.no_config5 and not args.generic and not c5_in_children:
        if ctx is not None:
            ctx.close()
            del x_dev, g_dev
        c5 = config5_block(args, rank, world, local_rank, rehearse)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_sweeps = B * args.steps * world
    value = total_sweeps / elapsed
    steps = max(args.steps, 1)
    # per-kernel rooflines; the one reported as `roofline` is the kernel with the longest launch (the dominant one)
    fwd_s, bwd_s = prof["fwd_ms"] / steps * 1e-3, prof["bwd_ms"] / steps * 1e-3
    en_s, gr_s = prof["energy_ms"] / steps * 1e-3, prof["grad_ms"] / steps * 1e-3
    alg_flop = B * (n_pts - 1) * 8.0 * d ** 3               # RK4: 4 stages x one D^3 product (symmetry) x 2 flop
    tj = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))

    traffic_missing = []

    def traffic_of(name):
        key = f"{name}_B{B}_D{d}_Np{n_pts}"
        if key not in tj:
            traffic_missing.append(key)              # (reported in the line: a null `traffic` is never silent)
        return tj.get(key)

    # NOT measured in this run: PMC counters need rocprofv3 passes of their own (tools/profile_bench.sh); the numbers are the
    # committed result of the last such run of this workload
    traffic_source = {"file": "profiles/pmc_traffic.json", "measured_in_this_run": False,
                      "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --no-single-problem "
                             "--no-cpu-baseline --no-config5` (tools/profile_bench.sh), bytes = 2 x FETCH_SIZE (gfx950 correction, "
                             "MI355X_MICROARCH.md HBM section) + WRITE_SIZE, per launch; raw counters: profiles/r05c_pmc_fetch_write_B512.csv"}

    # streams of the default batched path (33 <= D <= 40, RK2 / RK4, Sigma = sigma^2 I, B > #CUs): the backward kernel writes
    # Q''_t = A_t / sigma^2 - 2 Psi_t where Psi_t would be, the gradient assembly reads Q''_t and S_t only, and dEsde_dS exists as its
    # upper triangle (vgpa_hip.h VGPA_FLAG_KEEP_PSI; --keep-psi restores the round-2 streams)
    n_cu_ = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    sym_path = (d > 44 or (B > n_cu_ and (d + 3) // 4 <= 10) or ((d + 3) // 4 in (9, 10) and os.environ.get("VGPA_ODE_KERNEL") != "pe")) \
               and not args.generic and not args.keep_psi
    q_mode = sym_path and 33 <= d <= 40 and args.method.upper() in ("RK2", "RK4") and os.environ.get("VGPA_SYM_RUNS") != "1"
    tri = d * (d + 1) / 2.0 if sym_path else float(d * d)
    # ... and (round 4) S_t travels between the kernels as its packed lower triangle (OdeArgs::s_packed; VGPA_S_PACKED=0 keeps whole matrices)
    s_tri = d * (d + 1) / 2.0 if (q_mode and os.environ.get("VGPA_S_PACKED") != "0") else float(d * d)
    # ... and (round 5) from 64 problems on the backward RK4 kernel assembles the gradient itself on a third set of waves
    # (vgpa_api.hip::grad_fused_now): no Q'' stream, no assembly kernel -- three kernels per sweep
    fused = q_mode and args.method.upper() == "RK4" and s_tri < d * d and B >= 64 and os.environ.get("VGPA_FUSED_GRAD") != "0" \
        and os.environ.get("VGPA_SYM_COVER") != "op"
    grad_flop = B * n_pts * 2.0 * d ** 3                     # Q S per grid point
    kernels = {
        # stepping kernels: fp64 matrix pipe (AI = 8 D^3 / (16 D^2 ..) ~ D/2 flop/B > ridge ~10)
        "solve_fwd": dict(bound="mfma", seconds=fwd_s, alg=alg_flop, peak=FP64_PEAK_TFLOPS, scale=1e12, unit="TFLOP/s",
                          alg_bytes=B * 8.0 * n_pts * (d * d + s_tri + 2 * d)),   # read A,b ; write S (packed lower triangle), m
        "solve_bwd": dict(bound="mfma", seconds=bwd_s, alg=alg_flop, peak=FP64_PEAK_TFLOPS, scale=1e12, unit="TFLOP/s",
                          alg_bytes=B * 8.0 * n_pts * (2 * d * d + tri + 2 * d)),  # read A, dEsde/dS (upper), dEsde/dm ; write Psi | Q'', lam
        # the fused kernel: recursion + assembly; read A, dEsde/dS, dEsde/dm, S (packed), m, <f>, A m, b ; write gLa, gLb, lam
        "solve_bwd_grad": dict(bound="mfma", seconds=bwd_s, alg=alg_flop + grad_flop, peak=FP64_PEAK_TFLOPS, scale=1e12, unit="TFLOP/s",
                               alg_bytes=B * 8.0 * n_pts * (2 * d * d + tri + s_tri + 7 * d)),
        # per-grid-point kernels: HBM (energy: AI = 4 D^3 / (24 D^2) = D/6 flop/B; gradient: 2 D^3 / (32 D^2) = D/16)
        "energy_l96": dict(bound="hbm", seconds=en_s, alg=B * 8.0 * n_pts * (d * d + s_tri + tri + 6 * d), peak=HBM_PEAK_GBS, scale=1e9,
                           unit="GB/s"),                                          # read S, A, m, b ; write dEsde/dS (upper), dEsde/dm, <f>, A m, e_t
        "grad": dict(bound="hbm", seconds=gr_s, alg=B * 8.0 * n_pts * ((2 if q_mode else 3) * d * d + s_tri + 7 * d), peak=HBM_PEAK_GBS, scale=1e9,
                     unit="GB/s"),                                                # read Q'' (or A and Psi), S + vectors ; write gLa, gLb
    }
    if fused:
        del kernels["solve_bwd"], kernels["grad"]
    else:
        del kernels["solve_bwd_grad"]
    nb_blocks = (d + 3) // 4
    method_id = {"EULER": 0, "HEUN": 1, "RK2": 2, "RK4": 3}.get(args.method.upper(), 3)
    # device symbols as they appear in a rocprofv3 kernel trace (MFMA path, 5 <= D <= 64): symmetric-unit kernels from two
    # problems per CU on (and for 44 < D), role-specialised ones below (vgpa_api.hip::use_sym_units)
    n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    keep_pe = os.environ.get("VGPA_ODE_KERNEL") == "pe"
    sym_units = d > 44 or (B > n_cu and nb_blocks <= 10) or os.environ.get("VGPA_ODE_KERNEL") == "sym" or (nb_blocks in (9, 10) and not keep_pe)
    wpe = 2 if nb_blocks <= 10 else 1        # (helper-wave kernels with two helper roles: 3, see `hlp`)
    cover = 0 if (nb_blocks in (9, 10) and os.environ.get("VGPA_SYM_RUNS") != "1") else 1      # fragment-cover kernels for 33 <= D <= 40
    # (last parameter: the backward cover kernels of RK2 / RK4 store Q''_t = Sigma^-1 A_t - 2 Psi_t for the gradient assembly)
    q_out = lambda fwd: "true" if (fwd == "false" and cover == 0 and method_id in (2, 3) and not args.keep_psi) else "false"
    hlp = "true" if (cover == 0 and B <= n_cu and os.environ.get("VGPA_SYM_HELPERS") != "0") or os.environ.get("VGPA_SYM_HELPERS") == "1" else "false"
    step_sym = (lambda fwd: f"vgpa::sym::k_ode_sym<{method_id}, {fwd}, {nb_blocks}, false, {cover}, {wpe}, {q_out(fwd)}, 4, {hlp}, false, {hlp}>") if sym_units else \
               (lambda fwd: f"vgpa::mfma::k_ode_pe<{method_id}, {fwd}, {nb_blocks}, false>")
    symbols = {"solve_fwd": step_sym("true"), "solve_bwd": step_sym("false"),
               "solve_bwd_grad": f"vgpa::sym::k_ode_sym<{method_id}, false, {nb_blocks}, false, 0, 3, true, 4, true, true, false> (768 threads: product, helper and gradient waves)",
               "energy_l96": f"vgpa::(anonymous namespace)::k_energy_l96_r<{nb_blocks}, 1> (+ k_obs)", "grad": f"vgpa::k_grad_mfma{'_q' if (sym_units and q_out('false') == 'true') else ''}<{nb_blocks}> (+ k_reduce)"}
    roof = {}
    for name, k in kernels.items():
        ach = k["alg"] / max(k["seconds"], 1e-12) / k["scale"]
        roof[name] = {"bound": k["bound"], "kernel": name, "symbol": None if args.generic else symbols[name],
                      "achieved": ach, "peak": k["peak"], "unit": k["unit"],
                      "frac": ach / k["peak"], "traffic": traffic_of(name), "launch_ms": 1e3 * k["seconds"],
                      ("alg_flop_per_launch" if k["bound"] == "mfma" else "alg_bytes_per_launch"): k["alg"]}
    dom = max(kernels, key=lambda n: kernels[n]["seconds"])
    dom_s = kernels[dom]["seconds"]
    step_dom = ("solve_bwd_grad" if fused else "solve_bwd") if bwd_s >= fwd_s else "solve_fwd"
    alg_bytes = kernels[step_dom]["alg_bytes"]
    gbs = alg_bytes / kernels[step_dom]["seconds"] / 1e9
    sweep_bytes = 8.0 * n_pts * (5 * d * d + 6 * d)         # SURVEY.md s.8d algorithmic bytes of one fused sweep
    out = {
        "metric": "fwd+bwd sweeps/sec (free-energy+grad eval), Lorenz96 D=40 N=1000",
        "value": value, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (seeded Lorenz-96 path + observations, reference generator order)",
        "config": {"workload": f"Lorenz96 D={d}, {args.method.upper()}, Np={n_pts} (BASELINE configs[2])",
                   "batch_per_gpu": B, "sharding": "independent problems per GPU, no collective",
                   "kernels": "generic" if args.generic else "mfma"},
        # `value` is BATCH throughput (B independent sweeps per launch); ONE sweep -- what an SCG iteration of the reference's
        # use case waits for -- is latency-bound on one CU:
        "single_problem_sweeps_per_s": None if single is None else single["sweeps_per_s"],
        "whole_sweep_frac_of_hbm": sweep_bytes * value / world / 1e9 / HBM_PEAK_GBS,
        # the nominal 24 D^3 flop per grid point against the fp64 matrix peak: at D = 40 the sweep is bound by the
        # matrix pipe, not by HBM (19.6 us of fp64 work vs 8.2 us of HBM traffic per sweep), so THIS is the whole-sweep roofline
        "whole_sweep_frac_of_fp64": 24.0 * d ** 3 * n_pts * value / world / 1e12 / FP64_PEAK_TFLOPS,
        "roofline": dict(roof[dom], traffic_source=traffic_source, note=f"kernel with the longest launch of the sweep (batched, B={B}); energy+obs phase = "
                                         f"k_energy_l96_r + k_obs (0.1 ms); all {len(kernels)} kernels under roofline_kernels; whole "
                                         f"sweep = {sweep_bytes * value / world / 1e9 / HBM_PEAK_GBS:.3f} of HBM on SURVEY 8d's "
                                         f"algorithmic bytes; single problem = "
                                         f"{(single or {}).get('sweeps_per_s', float('nan')):.1f} sweeps/s"),
        # what `peak` is worth on this part (measured by tools/clock_under_load.sh + tools/sustained_peak.sh, parsed by tools/peak_context.py
        # into profiles/peak_context.json; not used for `frac`, not measured in this run)
        "peak_context": peak_context(),
        "roofline_kernels": roof,
        "traffic_missing": traffic_missing or None,      # keys profiles/pmc_traffic.json does not hold (tools/profile_bench.sh writes them)
        "roofline_hbm": {"bound": "hbm", "kernel": step_dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "alg_bytes_per_launch": alg_bytes,
                         "whole_sweep_GBs": sweep_bytes * value / world / 1e9,
                         "whole_sweep_frac": sweep_bytes * value / world / 1e9 / HBM_PEAK_GBS},
        "phase_ms_per_step": {"fwd": 1e3 * fwd_s, "energy+obs": 1e3 * en_s, "bwd": 1e3 * bwd_s,
                              "reduce+grad": prof["grad_ms"] / steps},
        "single_problem": single,
        "parity_check_rel_err_F": check,
        "parity_check_rel_err_grad_norm": check_g,
    }
    if c2 is not None:
        out["config2"] = c2
    if c4 is not None:
        out["config4"] = c4
    if c5 is not None:
        out["config5"] = c5

    if cpu is not None:                              # measured before the GPU work, rank 0 at N = 1 only
        f_cpu = cpu.pop("F_cpu")
        cpu["gpu_vs_cpu_F_rel_err"] = abs(np.atleast_1d(f_last)[0] - f_cpu) / abs(f_cpu)
        out["cpu_baseline"] = cpu
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
