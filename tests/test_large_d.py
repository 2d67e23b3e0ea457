"""
Large-D (D > 64) recursions.
  * GPU (-m gpu): the HIP stage kernels through the C ABI vs the numpy oracle, all four steppers, D in {80, 96, 128, 200}.
  * CPU: the row-sharding / collective logic of the host driver under gloo (world size 2), with a CPU stand-in for
    the two stage kernels that restates their contract (tests only -- the package ships only the HIP backend).
"""
import os
import sys
import socket

import numpy as np
import pytest

from conftest import ROOT, rel_err
from oracle import vgpa_oracle as vo

TOL = 1e-9


def make_inputs(d, n, seed=3):
    rng = np.random.default_rng(seed)
    a = 2.0 * np.eye(d) + 0.3 * rng.standard_normal((n, d, d)) / np.sqrt(d)
    b = rng.standard_normal((n, d))
    m0 = rng.standard_normal(d)
    q = rng.standard_normal((d, d)) / np.sqrt(d)
    s0 = 0.2 * np.eye(d) + 0.05 * (q + q.T)
    sigma = np.diag(1.0 + rng.random(d))
    g = rng.standard_normal((n, d, d)) / np.sqrt(d)
    gs = g + np.swapaxes(g, 1, 2)
    gm = rng.standard_normal((n, d))
    js, jm = np.zeros((n, d, d)), np.zeros((n, d))
    for t in range(3, n, 7):
        js[t] = 0.5 * np.eye(d)
        jm[t] = rng.standard_normal(d)
    return a, b, m0, s0, sigma, gm, gs, jm, js


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("d,n", [(80, 12), (96, 25), (128, 16), (200, 9)])
def test_large_d_matches_oracle(method, d, n):
    import vgpa_amd as va
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(d, n)
    mt, st = va.FwdOde(0.01, method, False)(a, b, m0, s0, sigma)
    mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, m0, s0, sigma)
    assert rel_err(mt, mt_o) < TOL and rel_err(st, st_o) < TOL
    lam, psi = va.BwdOde(0.01, method, False)(a, gm, gs, jm, js)
    lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, gs, jm, js)
    assert rel_err(lam, lam_o) < TOL and rel_err(psi, psi_o) < TOL


class _ThreadComm:
    """torch.distributed look-alike for `world` virtual ranks living in threads of ONE process that share one GPU:
    the collectives are barrier-ordered device copies.  Lets the row-sharded HIP kernels (row0 > 0, Mp < D, the packed
    column-chunk GEMM output) run on a one-GPU box; the RCCL calls themselves are covered by construction
    (same call sites) and by the gloo test below."""

    def __init__(self, world):
        import threading
        self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world

    def view(self, rank):
        return _RankComm(self, rank)


class _RankComm:
    def __init__(self, comm, rank):
        self.c, self.rank = comm, rank

    def get_world_size(self, group=None):
        return self.c.world

    def get_rank(self, group=None):
        return self.rank

    def _exchange(self, inp):
        import torch
        torch.cuda.synchronize()
        self.c.slots[self.rank] = inp
        self.c.barrier.wait()

    def _done(self):
        import torch
        torch.cuda.synchronize()
        self.c.barrier.wait()

    def all_to_all_single(self, out, inp, group=None):
        self._exchange(inp)
        n = inp.numel() // self.c.world
        for q in range(self.c.world):
            out[q * n:(q + 1) * n] = self.c.slots[q][self.rank * n:(self.rank + 1) * n]
        self._done()

    def all_gather_into_tensor(self, out, inp, group=None):
        self._exchange(inp)
        n = inp.numel()
        for q in range(self.c.world):
            if q != self.rank:
                out[q * n:(q + 1) * n] = self.c.slots[q]
        self._done()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("d,world", [(128, 2), (192, 4), (96, 3)])
def test_row_sharded_hip_kernels_with_virtual_ranks(method, d, world):
    """The sharded recursion on the real HIP kernels: every virtual rank must reproduce the unsharded oracle."""
    import threading
    import torch
    from legacy_sharded import ShardedRecursion
    n = 7
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(d, n)
    mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, m0, s0, sigma)
    lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, gs, jm, js)
    comm = _ThreadComm(world)
    errs, fails = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            rec = ShardedRecursion(method, 0.01, d, comm=comm.view(rank))
            assert rec.world == world and rec.Mp == d // world and rec.row0 == rank * (d // world)
            mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
            lam, psi = rec.solve_bwd(a, gm, gs, jm, js)
            torch.cuda.synchronize()
            errs[rank] = max(rel_err(mt.cpu().numpy(), mt_o), rel_err(st.cpu().numpy(), st_o),
                             rel_err(lam.cpu().numpy(), lam_o), rel_err(psi.cpu().numpy(), psi_o))
        except BaseException as exc:      # noqa: BLE001 - a dead thread would leave the others at the barrier
            fails.append(exc)
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not fails, fails
    assert all(e is not None and e < TOL for e in errs), errs


@pytest.mark.gpu
def test_large_d_agrees_with_small_d_kernels_at_the_boundary():
    """D = 64 runs on the LDS-resident kernels, the per-stage GEMM path must give the same numbers."""
    import vgpa_amd as va
    from legacy_sharded import ShardedRecursion
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(64, 20)
    mt, st = va.FwdOde(0.01, "rk4", False)(a, b, m0, s0, sigma)
    rec = ShardedRecursion("rk4", 0.01, 64)
    mt2, st2 = rec.solve_fwd(a, b, m0, s0, sigma)
    assert rel_err(mt2.cpu().numpy(), mt) < 1e-12 and rel_err(st2.cpu().numpy(), st) < 1e-12
    lam, psi = va.BwdOde(0.01, "rk4", False)(a, gm, gs, jm, js)
    lam2, psi2 = rec.solve_bwd(a, gm, gs, jm, js)
    assert rel_err(lam2.cpu().numpy(), lam) < 1e-12 and rel_err(psi2.cpu().numpy(), psi) < 1e-12


# ---------------------------------------------------------------------------------------------------------------
class CpuStageStandIn:
    """Restates the contract of vgpa_ld_gemm / vgpa_ld_stage (vgpa_amd/csrc/large_d.hip) with numpy -- tests only."""

    @staticmethod
    def _v(t, off, count):
        return t.numpy()[off:off + count]

    def gemm(self, transa, M, N, K, A0, a0_off, A1, a1_off, lda, B, ldb, C, cw):
        def op(t, off):
            flat = t.numpy()
            if transa:      # element (k, i) at off + k*lda + i
                return np.stack([flat[off + k * lda: off + k * lda + M] for k in range(K)]).T
            return np.stack([flat[off + i * lda: off + i * lda + K] for i in range(M)])
        a = op(A0, a0_off)
        if A1 is not None:
            a = 0.5 * (a + op(A1, a1_off))
        b = B.numpy()[:K * ldb].reshape(K, ldb)[:, :N]
        c = a.dot(b)
        out = C.numpy()
        for q in range(N // cw):
            out[q * M * cw:(q + 1) * M * cw] = c[:, q * cw:(q + 1) * cw].ravel()

    def stage(self, **kw):
        D, Mp, cw, row0 = kw["D"], kw["Mp"], kw["cw"], kw["row0"]
        fwd, ks, fin, cx, cf = kw["fwd"], kw["kstore"], kw["final_mode"], kw["cx"], kw["cf"]

        def mat(x, rows=Mp, cols=D):
            if x is None:
                return None
            t, off = x if isinstance(x, tuple) else (x, 0)
            return t.numpy()[off:off + rows * cols].reshape(rows, cols)

        def vec(x, count=Mp):
            if x is None:
                return None
            t, off = x if isinstance(x, tuple) else (x, 0)
            return t.numpy()[off:off + count]
        wp = kw["W"].numpy()[:(D // cw) * Mp * cw].reshape(D // cw, Mp, cw)
        w = np.concatenate([wp[q] for q in range(D // cw)], axis=1)
        wcol = kw["Wcol"].numpy()[:D * Mp].reshape(D, Mp) if kw["Wcol"] is not kw["W"] else w
        e = mat(kw["E0"]) if kw["E1"] is None else 0.5 * (mat(kw["E1"]) + mat(kw["E0"]))
        r = ((-w - wcol.T) + e) if fwd else ((-e + wcol.T) + w)
        k1, k23 = mat(kw["K1"]), mat(kw["K23"])
        sgn = 1.0 if fwd else -1.0
        base = mat(kw["base"]).copy()
        jump = mat(kw["J"]) if (fin and kw["J"] is not None) else 0.0
        k1_old, k23_old = k1.copy(), k23.copy()
        if ks == 1: k1[:] = r
        elif ks == 2: k23[:] = r
        elif ks == 3: k23[:] = k23 + r
        if fin == 0: res = base + sgn * (cx * r)
        else:
            comb = r if fin == 1 else ((k1_old + r) if fin == 2 else (k1_old + 2.0 * k23_old + r) / 6.0)
            res = base + sgn * (cf * comb) + jump
        mat(kw["out"])[:] = res
        # vector recursion
        a_t, a_off = kw["A0"]
        arows = a_t.numpy()[a_off + row0 * kw["lda"]: a_off + (row0 + Mp) * kw["lda"]].reshape(Mp, kw["lda"])[:, :D]
        if kw["A1"] is not None:
            b_t, b_off = kw["A1"]
            arows = 0.5 * (arows + b_t.numpy()[b_off + row0 * kw["lda"]: b_off + (row0 + Mp) * kw["lda"]].reshape(Mp, kw["lda"])[:, :D])
        y = arows.dot(kw["x"].numpy()[:D])
        ev = vec(kw["e0"]) if kw["e1"] is None else 0.5 * (vec(kw["e1"]) + vec(kw["e0"]))
        rv = (-y + ev) if fwd else (-ev + y)
        k1v, k23v = vec(kw["k1v"]), vec(kw["k23v"])
        k1o, k23o = k1v.copy(), k23v.copy()
        if ks == 1: k1v[:] = rv
        elif ks == 2: k23v[:] = rv
        elif ks == 3: k23v[:] = k23v + rv
        vb = vec(kw["vbase"]).copy()
        jv = vec(kw["jv"]) if (fin and kw["jv"] is not None) else 0.0
        if fin == 0: vres = vb + sgn * (cx * rv)
        else:
            comb = rv if fin == 1 else ((k1o + rv) if fin == 2 else (k1o + 2.0 * k23o + rv) / 6.0)
            vres = vb + sgn * (cf * comb) + jv
        vec(kw["vout"])[:] = vres


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir, method):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from vgpa_amd import parallel as par
    from legacy_sharded import ShardedRecursion
    par.init_from_env("gloo")
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(12, 7)
    rec = ShardedRecursion(method, 0.01, 12, backend=CpuStageStandIn())
    assert rec.world == world and rec.Mp == 12 // world
    mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
    lam, psi = rec.solve_bwd(a, gm, gs, jm, js)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), mt=mt.numpy(), st=st.numpy(), lam=lam.numpy(), psi=psi.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["rk4", "heun", "rk2", "euler"])
def test_row_sharded_recursion_two_ranks_gloo(tmp_path, method):
    """world = 2: all_to_all of the packed W chunks + all_gather of the row blocks reproduce the unsharded oracle."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), method), nprocs=2, join=True)
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(12, 7)
    mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, m0, s0, sigma)
    lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, gs, jm, js)
    for rank in range(2):
        z = np.load(tmp_path / f"r{rank}.npz")
        assert rel_err(z["mt"], mt_o) < 1e-12 and rel_err(z["st"], st_o) < 1e-12
        assert rel_err(z["lam"], lam_o) < 1e-12 and rel_err(z["psi"], psi_o) < 1e-12


def test_single_rank_driver_with_standin_matches_oracle():
    from legacy_sharded import ShardedRecursion
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(10, 6)
    for method in ("euler", "heun", "rk2", "rk4"):
        rec = ShardedRecursion(method, 0.01, 10, backend=CpuStageStandIn())
        mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
        mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, m0, s0, sigma)
        assert rel_err(mt.numpy(), mt_o) < 1e-12 and rel_err(st.numpy(), st_o) < 1e-12
        lam, psi = rec.solve_bwd(a, gm, gs, jm, js)
        lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, gs, jm, js)
        assert rel_err(lam.numpy(), lam_o) < 1e-12 and rel_err(psi.numpy(), psi_o) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("d,n", [(72, 9), (96, 8), (128, 7), (160, 6), (330, 5), (448, 4), (520, 3)])
def test_large_d_lorenz96_energy_terms(d, n):
    """vgpa_energy at D > 64 (batched blocked Cholesky / inverse / GEMMs) vs the oracle's lean L96 energy.  D = 330, 448, 520: six
    (the last one of 10 rows), seven and nine (8 rows) diagonal blocks -- every level of the inverse by halves has a ragged or a missing
    last group -- two to four segments of the residual sums, uneven halves of the batch on the two streams."""
    import vgpa_amd as va
    from test_gpu_edge_cases import make_problem, gpu_context
    p, x = make_problem("L96", d, n)
    a, b = p.split(x)
    mt, st = vo.solve_fwd(p.method, p.dt, False, a, b, p.m0, p.s0, p.sigma)
    esde_o, (ef_o, edf_o), (dm_o, ds_o, *_) = vo.model_energy(p, a, b, mt, st, faithful=False)
    ctx = va.Context("L96", "rk4", d, n, p.dt, sigma=p.sigma, theta=[8.0])
    esde, ef, edf, dm, ds = ctx.energy(a, b, mt, st)
    assert abs(esde - esde_o) <= TOL * abs(esde_o)
    assert rel_err(ef, ef_o) < TOL and rel_err(edf, edf_o) < TOL
    assert rel_err(dm, dm_o) < TOL and rel_err(ds, ds_o) < TOL


@pytest.mark.gpu
def test_energy_term_schedules_above_64_agree():
    """The round-5 schedule of lde_energy (two half-batches on two streams, inverse by halves, diagonal blocks on the matrix cores,
    XCD-balanced tile maps, mirrored dEsde_dS) against the round-4 one (VGPA_LDE_* switches, read once per process: a child process) at
    D = 330 and 192: the same energy terms to 1e-11, and dEsde_dS symmetric in every bit between different tiles (mirrored stores)."""
    import json
    import subprocess
    import vgpa_amd as va
    from test_gpu_edge_cases import make_problem
    cases = ((330, 5), (192, 4))
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import vgpa_amd as va\n"
        "from test_gpu_edge_cases import make_problem\n"
        "from oracle import vgpa_oracle as vo\n"
        "out = []\n"
        "for d, n in %r:\n"
        "    p, x = make_problem('L96', d, n)\n"
        "    a, b = p.split(x)\n"
        "    mt, st = vo.solve_fwd(p.method, p.dt, False, a, b, p.m0, p.s0, p.sigma)\n"
        "    ctx = va.Context('L96', 'rk4', d, n, p.dt, sigma=p.sigma, theta=[8.0])\n"
        "    esde, ef, edf, dm, ds = ctx.energy(a, b, mt, st)\n"
        "    out.append({'esde': float(esde), 'ef': np.asarray(ef).ravel().tolist(), 'dm': np.asarray(dm).ravel().tolist(),\n"
        "                'ds': np.asarray(ds).ravel().tolist()})\n"
        "    ctx.close()\n"
        "print(json.dumps(out))\n" % (os.path.dirname(__file__), cases))
    env = dict(os.environ)
    env.update({"VGPA_LDE_TWO_STREAMS": "0", "VGPA_LDE_INVERSE": "rows", "VGPA_LDE_DIAG": "valu", "VGPA_LDE_TILE_MAP": "0",
                "VGPA_LDE_SYRK_MIRROR": "0", "VGPA_LDE_PANEL": "1", "VGPA_LDE_K_DOWN": "0", "VGPA_LDE_GRAD_EPILOGUE": "0", "PYTHONPATH": os.pathsep.join([ROOT, env.get("PYTHONPATH", "")])})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    old = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("[")][-1])
    for (d, n), ref in zip(cases, old):
        p, x = make_problem("L96", d, n)
        a, b = p.split(x)
        mt, st = vo.solve_fwd(p.method, p.dt, False, a, b, p.m0, p.s0, p.sigma)
        ctx = va.Context("L96", "rk4", d, n, p.dt, sigma=p.sigma, theta=[8.0])
        esde, ef, edf, dm, ds = ctx.energy(a, b, mt, st)
        ctx.close()
        assert abs(esde - ref["esde"]) <= 1e-12 * abs(esde)
        assert rel_err(np.asarray(ef).ravel(), np.asarray(ref["ef"])) < 1e-12
        assert rel_err(np.asarray(dm).ravel(), np.asarray(ref["dm"])) < 1e-11
        assert rel_err(np.asarray(ds).ravel(), np.asarray(ref["ds"])) < 1e-11
        ds = np.asarray(ds).reshape(n, d, d)       # mirrored stores: symmetric in every bit between different 64 x 64 tiles
        blk = np.arange(d) // 64
        off = blk[:, None] != blk[None, :]
        assert np.array_equal(ds[:, off], ds.transpose(0, 2, 1)[:, off])


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,method", [(72, 9, "rk4"), (128, 7, "rk4"), (96, 8, "heun")])
def test_large_d_fused_sweep(d, n, method):
    """The whole sweep (F and gradient) at D > 64 against the oracle."""
    from test_gpu_edge_cases import make_problem, check
    p, x = make_problem("L96", d, n, method=method)
    check(p, x)


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["wide", "two-kernel"])
@pytest.mark.parametrize("d,n,method,batch", [(72, 9, "rk4", 1), (100, 7, "heun", 1), (130, 6, "rk2", 1), (96, 8, "euler", 1), (160, 6, "rk4", 1),
                                              (96, 7, "rk4", 3)])
def test_stage_kernel_versions_above_64(d, n, method, batch, version, monkeypatch):
    """The three implementations of a Runge-Kutta stage above D = 64 (large_d.hip: k_stage_prod up to D = 512 by default, k_stage_wide
    above, GEMM + k_stage_sym beyond that or on request) at the SAME small sizes -- ragged edge tiles, an odd number of diagonal
    tiles, a batch -- against the oracle.  VGPA_STAGE_FUSED / VGPA_STAGE_WIDE are read per call."""
    from test_gpu_edge_cases import make_problem, gpu_context
    monkeypatch.setenv("VGPA_STAGE_FUSED", "0")
    if version == "two-kernel":
        monkeypatch.setenv("VGPA_STAGE_WIDE", "0")
    p, x = make_problem("L96", d, n, method=method)
    ctx = gpu_context(p, batch=batch)
    rng = np.random.default_rng(5)
    xs = np.stack([x + 0.01 * rng.standard_normal(x.size) for _ in range(batch)])
    f, g = ctx.sweep(xs if batch > 1 else xs[0])
    f, g = np.atleast_1d(f), np.asarray(g).reshape(batch, -1)
    for q in range(batch):
        f_ref, g_ref, st = vo.sweep(p, xs[q], faithful=False)
        assert abs(f[q] - f_ref) <= TOL * abs(f_ref), (q, f[q], f_ref)
        assert rel_err(g[q], g_ref) < TOL, q
    for key in ("mt", "st", "lamt", "psit"):
        got = np.asarray(ctx.fetch(key))
        got = got[batch - 1] if batch > 1 else got
        assert rel_err(got.reshape(np.shape(st[key])), st[key]) < TOL, key
    s_t = np.asarray(ctx.fetch("st"))
    s_t = s_t[0] if batch > 1 else s_t
    assert np.array_equal(s_t, np.swapaxes(s_t, -1, -2))          # exactly symmetric S_t (pairs and diagonal tiles alike)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,method", [(544, 4, "rk4"), (530, 4, "heun"), (578, 4, "rk2")])
def test_throughput_stage_kernel_above_512(d, n, method):
    """k_stage_wide at its own sizes: 17 tiles (an odd number of diagonal tiles), ragged edge tiles (530 = 16 x 32 + 18)."""
    from test_gpu_edge_cases import make_problem, check
    p, x = make_problem("L96", d, n, method=method)
    check(p, x)


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,method,batch", [(72, 9, "rk4", 1), (128, 7, "heun", 1), (96, 8, "rk2", 1), (80, 11, "euler", 1), (96, 9, "rk4", 3)])
def test_repeated_sweeps_on_one_context_above_64(d, n, method, batch):
    """An optimiser's use of a context above D = 64: sweep after sweep on the same device buffers, every one on another
    parameter vector and compared with the oracle; the stored recursions fetched in between; F-only evaluations followed by
    the gradient (all four steppers through the one-kernel stages, one batched context)."""
    from test_gpu_edge_cases import make_problem, gpu_context
    p, x = make_problem("L96", d, n, method=method)
    ctx = gpu_context(p, batch=batch)
    rng = np.random.default_rng(11)
    for it in range(5):
        xs = np.stack([x + 0.01 * rng.standard_normal(x.size) for _ in range(batch)])
        f, g = ctx.sweep(xs if batch > 1 else xs[0])
        f, g = np.atleast_1d(f), np.asarray(g).reshape(batch, -1)
        for q in range(batch):
            f_ref, g_ref, st = vo.sweep(p, xs[q], faithful=False)
            assert abs(f[q] - f_ref) <= TOL * abs(f_ref), (it, q, f[q], f_ref)
            assert rel_err(g[q], g_ref) < TOL, (it, q)
        if it in (2, 4):
            for key in ("mt", "st", "lamt", "psit"):
                got = np.asarray(ctx.fetch(key))
                got = got[batch - 1] if batch > 1 else got
                assert rel_err(got.reshape(np.shape(st[key])), st[key]) < TOL, (it, key)
    if batch == 1:                                  # F-only evaluations, the gradient asked for afterwards
        for it in range(3):
            xs = x + 0.01 * rng.standard_normal(x.size)
            f = ctx.free_energy(xs)
            f_ref, g_ref, _ = vo.sweep(p, xs, faithful=False)
            assert abs(float(np.atleast_1d(f)[0]) - f_ref) <= TOL * abs(f_ref)
            assert rel_err(ctx.gradient(), g_ref) < TOL
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("d,n,chunk", [(80, 23, 5), (96, 12, 11), (72, 30, 1), (128, 9, 64)])
def test_time_chunked_sweep_equals_the_resident_one(d, n, chunk, method):
    """VGPA_FLAG_STREAM_LARGE_D keeps only x, S_t and the gradient resident (BASELINE config 4 does not fit otherwise):
    F, the gradient and the state that is still kept must equal the resident sweep's and the oracle's; chunk sizes
    that do / do not divide the grid, a single-point chunk and one chunk for the whole grid."""
    from test_gpu_edge_cases import make_problem, gpu_context
    from vgpa_amd._lib import FLAG_STREAM_LARGE_D, OPT_LD_CHUNK
    p, x = make_problem("L96", d, n, method=method)
    res = gpu_context(p)
    assert not res.streaming
    f_r, g_r = res.sweep(x)
    st = gpu_context(p, flags=FLAG_STREAM_LARGE_D)
    assert st.streaming
    st.set_option(OPT_LD_CHUNK, chunk)
    f_s, g_s = st.sweep(x)
    assert f_s == f_r and np.array_equal(g_s, g_r)              # same kernels, same order per grid point
    f_o, g_o, state = vo.sweep(p, x, faithful=False)
    assert abs(f_s - f_o) <= TOL * abs(f_o) and rel_err(g_s, g_o) < TOL
    # split evaluation: free_energy (no backward pass needed for F), then the gradient from the cached (m, S)
    assert st.free_energy(x) == f_r
    assert np.array_equal(st.gradient(None), g_r)
    for key in ("mt", "st", "lamt"):
        assert np.array_equal(st.fetch(key), res.fetch(key)), key
    with pytest.raises(NotImplementedError):
        st.fetch("psit")
    # device-resident entry points
    xb, gb = st.alloc(x.size), st.alloc(x.size)
    xb.upload(x)
    assert st.sweep_dev(xb, gb) == f_r and np.array_equal(gb.download(), g_r)
    with pytest.raises(RuntimeError):
        st.set_option(OPT_LD_CHUNK, 3)                          # buffers exist already
    st.close(); res.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
def test_library_gemm_backend_of_the_stage_product(method):
    """VGPA_FLAG_LIBRARY_GEMM swaps the plain stage GEMM for rocBLAS dgemm (dlopen'ed); same sweep to 1e-9 of the oracle
    and of the hand-written GEMM, resident and time-chunked."""
    from test_gpu_edge_cases import make_problem, gpu_context
    from vgpa_amd._lib import FLAG_LIBRARY_GEMM, FLAG_STREAM_LARGE_D
    p, x = make_problem("L96", 96, 14, method=method)
    f_o, g_o, _ = vo.sweep(p, x, faithful=False)
    own = gpu_context(p)
    f_h, g_h = own.sweep(x)
    for flags in (FLAG_LIBRARY_GEMM, FLAG_LIBRARY_GEMM | FLAG_STREAM_LARGE_D):
        ctx = gpu_context(p, flags=flags)
        f, g = ctx.sweep(x)
        assert abs(f - f_o) <= TOL * abs(f_o) and rel_err(g, g_o) < TOL
        assert abs(f - f_h) <= 1e-12 * abs(f_h) and rel_err(g, g_h) < 1e-11
        ctx.close()
    own.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,method", [(96, 9, "rk4"), (128, 8, "heun"), (256, 6, "rk4"), (80, 7, "rk2"), (72, 9, "euler")])
def test_batched_contexts_above_64(d, n, method):
    """64 < D <= 512 with batch = 4 (one problem of that size leaves the chip idle and the reference treats every D
    alike, src/numerics/ode_solver.py:31-95): the per-stage GEMM / stage kernels take the problems in grid.z.  Every problem's F,
    gradient and state arrays against the oracle; a batch equals the single-problem contexts bit for bit."""
    from test_gpu_edge_cases import make_problem, gpu_context
    p, x = make_problem("L96", d, n, method=method)
    rng = np.random.default_rng(d + n)
    xb = np.stack([x + 0.01 * rng.standard_normal(x.size) for _ in range(4)])
    ctx = gpu_context(p, batch=4)
    fb, gb = ctx.sweep(xb)
    mt_b, st_b, psi_b = ctx.fetch("mt"), ctx.fetch("st"), ctx.fetch("psit")
    one = gpu_context(p)
    for i in range(4):
        f_o, g_o, st_o = vo.sweep(p, xb[i], faithful=False)
        assert abs(fb[i] - f_o) <= TOL * abs(f_o), (i, fb[i], f_o)
        assert rel_err(gb[i], g_o) < TOL
        assert rel_err(np.asarray(mt_b)[i], st_o["mt"]) < TOL and rel_err(np.asarray(st_b)[i], st_o["st"]) < TOL
        assert rel_err(np.asarray(psi_b)[i], st_o["psit"]) < TOL
        f1, g1 = one.sweep(xb[i])
        assert f1 == fb[i] and np.array_equal(g1, gb[i])
    ctx.close(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d,method", [(72, "rk4"), (96, "heun"), (128, "rk4")])
def test_dense_noise_matrices_above_64(d, method):
    """D > 64 with a DENSE system noise Sigma (the gradient takes any Sigma^-1, variational.py:320,332: one more batched product),
    dense S0, dense observation noise R and a non-identity observation operator H: fused sweep and state arrays vs the oracle."""
    from test_gpu_edge_cases import make_problem, gpu_context
    rng = np.random.default_rng(d)
    h = np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
    p, x = make_problem("L96", d, 8, method=method, dense=True, h_op=h)
    f_o, g_o, st_o = vo.sweep(p, x, faithful=False)
    ctx = gpu_context(p)
    f, g = ctx.sweep(x)
    assert abs(f - f_o) <= TOL * abs(f_o), (f, f_o)
    assert rel_err(g, g_o) < TOL
    for key in ("mt", "st", "lamt", "psit"):
        assert rel_err(np.asarray(ctx.fetch(key)).reshape(np.shape(st_o[key])), st_o[key]) < TOL, key
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d", [72, 100])
def test_hyper_parameter_members_above_64(d):
    """dEsde_dth / dEsde_dSig of Lorenz96.energy (lorenz_96.py:421-434: computed by the reference, consumed by nothing) at D > 64,
    through the mirror of the reference's model class."""
    from helpers import make_model
    from test_gpu_edge_cases import make_problem
    p, x = make_problem("L96", d, 7, method="rk4")
    a, b = p.split(x)
    mt, st = vo.solve_fwd("rk4", p.dt, False, a, b, p.m0, p.s0, p.sigma)
    want = vo.energy_l96(p.theta, p.inverse_sigma, p.dt, a, b, mt, st, list(p.obs_t), faithful=False)
    model = make_model("L96", d)
    model.sigma = p.sigma
    model.sample_path, model.time_window = np.zeros((7, d)), np.arange(7) * p.dt
    esde, (ef, edf), (dm, ds, dth, dsig) = model.energy(a, b, mt, st, list(p.obs_t))
    assert abs(esde - want[0]) <= TOL * abs(want[0])
    assert rel_err(ef, want[1][0]) < TOL and rel_err(dm, want[2][0]) < TOL and rel_err(ds, want[2][1]) < TOL
    assert rel_err(np.asarray(dth), want[2][2]) < TOL
    assert rel_err(np.asarray(dsig), want[2][3]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("d", [80, 130])
def test_non_symmetric_operator_inputs_above_64(d, method):
    """Operator-level calls with a non-symmetric S0 / dEsde_dS / matrix jump at D > 64 follow A.S + S.A^T and Psi.A + A^T.Psi
    literally (ode_solver.py:60,94) -- no symmetry shortcut, like the generic kernels at D <= 64."""
    import vgpa_amd as va
    rng = np.random.default_rng(d)
    n = 9
    a = 2.0 * np.eye(d) + 0.3 * rng.standard_normal((n, d, d)) / np.sqrt(d)
    b = rng.standard_normal((n, d))
    s0 = 0.2 * np.eye(d) + 0.02 * rng.standard_normal((d, d)) / np.sqrt(d)
    sigma = np.diag(1.0 + rng.random(d))
    mt, st = va.FwdOde(0.01, method, False)(a, b, np.zeros(d), s0, sigma)
    mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, np.zeros(d), s0, sigma)
    assert rel_err(mt, mt_o) < TOL and rel_err(st, st_o) < TOL
    assert rel_err(st, np.swapaxes(st, 1, 2)) > 1e-6                     # the result really is non-symmetric
    g = rng.standard_normal((n, d, d)) / np.sqrt(d)
    js = np.zeros((n, d, d)); js[4] = rng.standard_normal((d, d)) / np.sqrt(d)
    gm, jm = rng.standard_normal((n, d)), np.zeros((n, d))
    lam, psi = va.BwdOde(0.01, method, False)(a, gm, g, jm, js)
    lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, g, jm, js)
    assert rel_err(lam, lam_o) < TOL and rel_err(psi, psi_o) < TOL


@pytest.mark.gpu
def test_batched_operators_above_64():
    """Operator-level calls (FwdOde / BwdOde contract) with three problems at D = 96: problem-major inputs, every problem
    against the oracle."""
    import vgpa_amd as va
    d, n, nb = 96, 7, 3
    sets = [make_inputs(d, n, seed=11 + i) for i in range(nb)]
    ctx = va.Context("NONE", "rk4", d, n, 0.01, sigma=sets[0][4], batch=nb)
    a = np.stack([s_[0] for s_ in sets]); b = np.stack([s_[1] for s_ in sets])
    mt, st = ctx.solve_fwd(a, b, sets[0][2], sets[0][3], sets[0][4])
    gm = np.stack([s_[5] for s_ in sets]); gs = np.stack([s_[6] for s_ in sets])
    jm = np.stack([s_[7] for s_ in sets]); js = np.stack([s_[8] for s_ in sets])
    lam, psi = ctx.solve_bwd(a, gm, gs, jm, js)
    for i in range(nb):
        mt_o, st_o = vo.solve_fwd("rk4", 0.01, False, a[i], b[i], sets[0][2], sets[0][3], sets[0][4])
        lam_o, psi_o = vo.solve_bwd("rk4", 0.01, False, a[i], gm[i], gs[i], jm[i], js[i])
        assert rel_err(mt[i], mt_o) < TOL and rel_err(st[i], st_o) < TOL
        assert rel_err(lam[i], lam_o) < TOL and rel_err(psi[i], psi_o) < TOL
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------
# The native driver: step / stage loop and collectives inside libvgpa_hip.so (vgpa_shard_*).
class _CallbackComm:
    """`world` virtual ranks in threads of ONE process sharing one GPU: a vgpa_comm table whose collectives are
    barrier-ordered device copies (hipMemcpyAsync through ctypes).  Exercises the C++ driver's schedule, pointer
    arithmetic, in-place gathers, the point-to-point groups of the pipelined gather (send / recv between group_begin and
    group_end, matched per peer in posting order like ncclSend / ncclRecv) and the failure path (`abort` breaks the barrier,
    so every rank's next collective fails -- what ncclCommAbort plus the shard's bounded waits do on real hardware).
    `fail_at = (rank, n)`: that rank's n-th table call returns an error instead of taking part.
    `p2p=False` hands out a table without send / recv: the driver then keeps the serial schedule.
    RCCL itself sits behind the same table (tests/test_large_d.py::test_rccl_*)."""

    def __init__(self, world, fail_at=None, p2p=True):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.send = [None] * world
        self.posted = [None] * world           # per rank: the sends of its open group, [(peer, ptr, count)]
        self.hip = ctypes_hip()
        self.fail_at, self.p2p = fail_at, p2p
        self.calls = [0] * world
        self.p2p_groups = [0] * world          # groups that carried point-to-point traffic (the pipelined gather ran)

    def table(self, rank):
        import ctypes
        from vgpa_amd._lib import VgpaComm, COMM_COLLECTIVE, COMM_GROUP, COMM_P2P
        hip, world = self.hip, self.world
        group = {"open": False, "sends": [], "recvs": [], "stream": None}

        def guarded(fn):
            def call(*a):
                try:
                    self.calls[rank] += 1
                    if self.fail_at is not None and self.fail_at == (rank, self.calls[rank]):
                        return 7                                  # this rank's collective fails; it never reaches the barrier
                    return fn(*a)
                except BaseException:                             # noqa: BLE001 - a broken barrier is a failed collective
                    return 9
            return call

        def wait():
            self.barrier.wait(timeout=300)

        def publish(send, stream):
            hip.hipStreamSynchronize(ctypes.c_void_p(stream))
            self.send[rank] = send
            wait()

        def finish(stream):
            hip.hipStreamSynchronize(ctypes.c_void_p(stream))
            wait()

        def all_gather(user, send, recv, count, stream):
            publish(send, stream)
            for q in range(world):
                if q != rank or send != recv + rank * count * 8:
                    hip.hipMemcpyAsync(ctypes.c_void_p(recv + q * count * 8), ctypes.c_void_p(self.send[q]),
                                       ctypes.c_size_t(count * 8), 3, ctypes.c_void_p(stream))
            finish(stream)
            return 0

        def all_to_all(user, send, recv, count, stream):
            publish(send, stream)
            for q in range(world):
                hip.hipMemcpyAsync(ctypes.c_void_p(recv + q * count * 8), ctypes.c_void_p(self.send[q] + rank * count * 8),
                                   ctypes.c_size_t(count * 8), 3, ctypes.c_void_p(stream))
            finish(stream)
            return 0

        def group_begin(user):
            group.update(open=True, sends=[], recvs=[], stream=None)
            return 0

        def send(user, buf, count, peer, stream):
            assert group["open"] and 0 <= peer < world and peer != rank
            group["sends"].append((peer, buf, count)); group["stream"] = stream
            return 0

        def recv(user, buf, count, peer, stream):
            assert group["open"] and 0 <= peer < world and peer != rank
            group["recvs"].append((peer, buf, count)); group["stream"] = stream
            return 0

        def group_end(user):
            group["open"] = False
            if not group["sends"] and not group["recvs"]:
                return 0
            stream = group["stream"]
            hip.hipStreamSynchronize(ctypes.c_void_p(stream))
            self.posted[rank] = list(group["sends"])
            wait()
            taken = {}
            for peer, buf, count in group["recvs"]:               # the k-th receive from a peer meets its k-th send to this rank
                mine = [t for t in self.posted[peer] if t[0] == rank]
                k = taken.get(peer, 0)
                _, src, n = mine[k]
                assert n == count, (n, count)
                taken[peer] = k + 1
                hip.hipMemcpyAsync(ctypes.c_void_p(buf), ctypes.c_void_p(src), ctypes.c_size_t(count * 8), 3, ctypes.c_void_p(stream))
            self.p2p_groups[rank] += 1
            finish(stream)
            return 0

        def abort(user):
            self.barrier.abort()
            return 0

        t = VgpaComm()
        t._keep = (COMM_COLLECTIVE(guarded(all_gather)), COMM_COLLECTIVE(guarded(all_to_all)), COMM_GROUP(guarded(group_begin)),
                   COMM_GROUP(guarded(group_end)), COMM_P2P(guarded(send)), COMM_P2P(guarded(recv)), COMM_GROUP(abort))      # keep the thunks alive
        t.user, t.all_gather, t.all_to_all, t.group_begin, t.group_end = None, t._keep[0], t._keep[1], t._keep[2], t._keep[3]
        if self.p2p:
            t.send, t.recv = t._keep[4], t._keep[5]
        else:
            t.send, t.recv = COMM_P2P(), COMM_P2P()
        t.abort = t._keep[6]
        return t


def ctypes_hip():
    """ctypes handle of the HIP runtime THIS process already uses (torch bundles its own copy: a second one would not
    know the streams of the first)."""
    import ctypes
    import torch  # noqa: F401
    import vgpa_amd
    vgpa_amd.load()
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    if not paths:
        raise RuntimeError("HIP runtime not loaded")
    return ctypes.CDLL(paths[0])


def _run_native_virtual_ranks(method, d, n, world, chunks=None):
    """chunks: None = the library's default schedule (pipelined gather wherever the row block allows it), 0 = the serial
    schedule, k = k sub-blocks."""
    import threading
    import torch
    from vgpa_amd.large_d import NativeShardedRecursion
    from vgpa_amd._lib import SHARD_OPT_GATHER_CHUNKS
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(d, n)
    mt_o, st_o = vo.solve_fwd(method, 0.01, False, a, b, m0, s0, sigma)
    lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a, gm, gs, jm, js)
    comm = _CallbackComm(world)
    errs, fails, slices, used = [None] * world, [], [None] * world, [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)
            rec = NativeShardedRecursion(method, 0.01, d, n, rank=rank, world=world, device=0, comm=comm.table(rank))
            if chunks is not None:
                rec.set_option(SHARD_OPT_GATHER_CHUNKS, chunks)
            used[rank] = rec.gather_chunks
            lo, hi = rec.time_slice
            slices[rank] = (lo, hi)
            mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
            lam, psi = rec.solve_bwd(a, gm, gs, jm, js)
            e = 0.0
            if hi > lo:
                e = max(rel_err(mt.cpu().numpy(), mt_o[lo:hi]), rel_err(st.cpu().numpy(), st_o[lo:hi]),
                        rel_err(lam.cpu().numpy(), lam_o[lo:hi]), rel_err(psi.cpu().numpy(), psi_o[lo:hi]))
            errs[rank] = e
            rec.close()
        except BaseException as exc:      # noqa: BLE001 - a dead thread would leave the others at the barrier
            fails.append(exc)
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not fails, fails
    assert all(e is not None and e < TOL for e in errs), errs
    # the owners' slices tile the grid
    assert slices[0][0] == 0 and slices[-1][1] == n and all(slices[r][1] == slices[r + 1][0] for r in range(world - 1))
    # the schedule that ran: the same on every rank, and the pipelined one really went through the point-to-point groups
    assert len(set(used)) == 1
    if world > 1:
        assert (comm.p2p_groups[0] > 0) == (used[0] > 0)
    return used[0]


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("d,n,world", [(128, 7, 2), (192, 6, 4), (96, 5, 3), (80, 9, 1)])
def test_native_sharded_driver_with_virtual_ranks(method, d, n, world):
    """vgpa_shard_solve_fwd / _bwd (C++ step / stage loop + collectives through vgpa_comm): every virtual rank's time slice
    must equal the unsharded oracle; also with more ranks than some slices have grid points."""
    used = _run_native_virtual_ranks(method, d, n, world)
    # the default schedule is the pipelined gather wherever a sub-block can hold whole 16-row k-tiles
    assert used == {(128, 2): 4, (192, 4): 3, (96, 3): 2, (80, 1): 0}[(d, world)]


@pytest.mark.gpu
@pytest.mark.parametrize("method,d,n,world,chunks", [("rk4", 128, 7, 2, 0), ("rk4", 128, 7, 2, 1), ("rk4", 128, 7, 2, 2),
                                                     ("rk2", 192, 6, 4, 0), ("heun", 192, 6, 4, 1), ("euler", 256, 5, 4, 4),
                                                     ("rk4", 256, 5, 2, 8)])
def test_native_sharded_driver_schedules(method, d, n, world, chunks):
    """The serial schedule (one grouped all-gather per stage on the compute stream) and the pipelined one with 1, 2, 4, 8
    sub-blocks (second stream, per-sub-block events, K-chunk launches of the next product): same recursion, same oracle."""
    used = _run_native_virtual_ranks(method, d, n, world, chunks=chunks)
    assert used == chunks


@pytest.mark.gpu
def test_native_sharded_driver_without_point_to_point_table_stays_serial():
    """A vgpa_comm table without send / recv (round 2's layout): the driver keeps the serial schedule."""
    import threading
    import torch
    from vgpa_amd.large_d import NativeShardedRecursion
    d, n, world = 128, 5, 2
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(d, n)
    mt_o, st_o = vo.solve_fwd("rk4", 0.01, False, a, b, m0, s0, sigma)
    comm = _CallbackComm(world, p2p=False)
    errs, fails = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            rec = NativeShardedRecursion("rk4", 0.01, d, n, rank=rank, world=world, device=0, comm=comm.table(rank))
            assert rec.gather_chunks == 0
            lo, hi = rec.time_slice
            mt, st = rec.solve_fwd(a, b, m0, s0, sigma)
            errs[rank] = max(rel_err(mt.cpu().numpy(), mt_o[lo:hi]), rel_err(st.cpu().numpy(), st_o[lo:hi]))
            rec.close()
        except BaseException as exc:      # noqa: BLE001
            fails.append(exc)
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not fails, fails
    assert all(e is not None and e < TOL for e in errs) and comm.p2p_groups == [0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("transa", [False, True])
@pytest.mark.parametrize("M,D,world,chunks", [(512, 4096, 8, 4), (128, 1024, 8, 2), (64, 128, 2, 4), (32, 96, 3, 2)])
def test_stage_gemm_k_chunk_launches(M, D, world, chunks, transa):
    """vgpa_ld_gemm_chunk -- the K-chunk launches of the pipelined stage product: launch j multiplies sub-block j of every
    rank's row block of the operand (k-tiles segmented with stride Mp) and continues the sums stored in W.  All launches
    together = the whole product (numpy, 1e-12); one launch over every k in natural order = vgpa_ld_gemm bit for bit (the
    accumulators continue one fp64 chain).  BASELINE configs[4]'s row block (512 x 4096 x 4096, 8 ranks, 4 sub-blocks: the
    16-byte-load kernel) down to shapes that take the bounds-checked kernel."""
    import ctypes
    import torch
    from legacy_sharded import HipStageBackend
    be = HipStageBackend()
    lib, st = be._lib, be._stream()
    rng = np.random.default_rng(M + D + world)
    a_h = rng.standard_normal((D, M) if transa else (M, D))          # op(A) = [M][K = D]
    x_h = rng.standard_normal((D, D))
    dev = torch.device("cuda", 0)
    a_d, x_d = torch.as_tensor(a_h, device=dev), torch.as_tensor(x_h, device=dev)
    w_d = torch.full((M * D,), float("nan"), dtype=torch.float64, device=dev)
    mp = D // world
    sub = mp // chunks
    lda = M if transa else D
    for j in range(chunks):
        a_off = j * sub * lda if transa else j * sub
        rc = lib.vgpa_ld_gemm_chunk(st, int(transa), M, D, world * sub, be._p(a_d, a_off), lda, be._p(x_d, j * sub * D), D, be._p(w_d), D,
                                    sub // 16, mp, int(j > 0))
        assert rc == 0
    torch.cuda.synchronize()
    want = (a_h.T if transa else a_h).dot(x_h)
    assert rel_err(w_d.cpu().numpy().reshape(M, D), want) < 1e-12
    # one segmented launch over all of K in natural order continues nothing and skips nothing: the plain product, bit for bit
    w1, w2 = torch.zeros_like(w_d), torch.zeros_like(w_d)
    assert lib.vgpa_ld_gemm_chunk(st, int(transa), M, D, D, be._p(a_d), lda, be._p(x_d), D, be._p(w1), D, mp // 16, mp, 0) == 0
    assert lib.vgpa_ld_gemm(st, int(transa), M, D, D, be._p(a_d), None, lda, be._p(x_d), D, be._p(w2), D) == 0
    torch.cuda.synchronize()
    assert torch.equal(w1, w2)
    assert lib.vgpa_ld_gemm_chunk(st, int(transa), M, D, D, be._p(a_d), lda, be._p(x_d), D, be._p(w1), D, 3, 16, 0) == -1   # stride < segment


@pytest.mark.gpu
@pytest.mark.parametrize("method,d,n,world", [("rk4", 128, 11, 2), ("rk4", 96, 13, 3), ("heun", 192, 9, 4), ("euler", 80, 7, 1),
                                              ("rk2", 128, 10, 4), ("rk4", 1024, 10, 8)])
def test_native_sharded_fused_sweep_with_virtual_ranks(method, d, n, world):
    """vgpa_shard_sweep: forward (row-sharded) -> time-parallel observation / E_sde terms -> one all-gather -> backward
    (row-sharded) -> time-parallel gradient.  F on every rank and every rank's gradient slice vs the oracle's sweep."""
    import threading
    import torch
    from vgpa_amd.large_d import NativeShardedRecursion
    from test_gpu_edge_cases import make_problem
    p, x = make_problem("L96", d, n, method=method)
    f_o, g_o, _ = vo.sweep(p, x, faithful=False)
    ga_o, gb_o = g_o[:n * d * d].reshape(n, d, d), g_o[n * d * d:].reshape(n, d)
    e0 = float(np.asarray(vo.kl0(p)))
    comm = _CallbackComm(world)
    errs, fails = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            rec = NativeShardedRecursion(method, p.dt, d, n, rank=rank, world=world, device=0,
                                         comm=comm.table(rank) if world > 1 else None)
            lo, hi = rec.time_slice
            for _ in range(2):             # twice: the buffers of the first call are reused
                f, ga, gb = rec.sweep(x, p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), e0)
            e = abs(f - f_o) / abs(f_o)
            if hi > lo:
                e = max(e, rel_err(ga.cpu().numpy(), ga_o[lo:hi]), rel_err(gb.cpu().numpy(), gb_o[lo:hi]))
            errs[rank] = e
            rec.close()
        except BaseException as exc:      # noqa: BLE001 - a dead thread would leave the others at the barrier
            fails.append(exc)
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not fails, fails
    assert all(e is not None and e < TOL for e in errs), errs


def _virtual_ranks(world, body, comm=None):
    """Runs body(rank, comm) in `world` threads; returns (results, exceptions)."""
    import threading
    import torch
    comm = comm if comm is not None else _CallbackComm(world)
    out, fails = [None] * world, [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)
            out[rank] = body(rank, comm)
        except BaseException as exc:      # noqa: BLE001 - a dead thread would leave the others at the barrier
            fails[rank] = exc
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    return out, fails, comm


@pytest.mark.gpu
@pytest.mark.parametrize("method,d,n,world,chunks", [("rk4", 128, 11, 2, None), ("rk4", 96, 13, 3, None), ("heun", 192, 9, 4, 0),
                                                     ("rk2", 128, 10, 4, None), ("euler", 80, 7, 1, None), ("rk4", 1024, 10, 8, None)])
def test_native_sharded_sweep_with_memory_sharded_x(method, d, n, world, chunks):
    """vgpa_shard_sweep_sharded: x handed over TIME-sharded like the gradient (a_own / b_own: the rank's grid points only).
    Rows (forward) and columns (backward) of every A_t reach the row-sharded recursions through time -> block exchanges, dEsde_dS
    through a time -> row exchange: no rank holds a complete (Np, D, D) array.  F and every gradient slice vs the oracle; slices of
    uneven length and ranks with more grid points than an exchange batch included."""
    from vgpa_amd.large_d import NativeShardedRecursion
    from vgpa_amd._lib import SHARD_OPT_GATHER_CHUNKS
    from test_gpu_edge_cases import make_problem
    p, x = make_problem("L96", d, n, method=method)
    f_o, g_o, _ = vo.sweep(p, x, faithful=False)
    ga_o, gb_o = g_o[:n * d * d].reshape(n, d, d), g_o[n * d * d:].reshape(n, d)
    a_h, b_h = x[:n * d * d].reshape(n, d, d), x[n * d * d:].reshape(n, d)
    e0 = float(np.asarray(vo.kl0(p)))

    def body(rank, comm):
        rec = NativeShardedRecursion(method, p.dt, d, n, rank=rank, world=world, device=0, comm=comm.table(rank) if world > 1 else None)
        if chunks is not None:
            rec.set_option(SHARD_OPT_GATHER_CHUNKS, chunks)
        lo, hi = rec.time_slice
        for _ in range(2):             # twice: the buffers of the first call are reused
            f, ga, gb = rec.sweep_sharded(a_h[lo:hi], b_h[lo:hi], p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y,
                                          np.diag(p.obs_noise), e0)
        e = abs(f - f_o) / abs(f_o)
        if hi > lo:
            e = max(e, rel_err(ga.cpu().numpy(), ga_o[lo:hi]), rel_err(gb.cpu().numpy(), gb_o[lo:hi]))
        # as an optimisation calls it: the fixed operands prepared once, the gradient pair of the sweep before written again
        problem = rec.prepare(p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), e0)
        first = (ga.cpu().numpy(), gb.cpu().numpy())
        f2, ga2, gb2 = rec.sweep_sharded(a_h[lo:hi], 1.01 * b_h[lo:hi], problem, out=(ga, gb))     # another x: the pair is overwritten
        assert ga2.data_ptr() == ga.data_ptr() and gb2.data_ptr() == gb.data_ptr() and f2 != f
        f3, ga3, gb3 = rec.sweep_sharded(a_h[lo:hi], b_h[lo:hi], problem, out=(ga2, gb2))
        assert abs(f3 - f) <= 1e-12 * abs(f) and ga3.data_ptr() == ga.data_ptr()
        if hi > lo:
            assert rel_err(ga3.cpu().numpy(), first[0]) < 1e-12 and rel_err(gb3.cpu().numpy(), first[1]) < 1e-12
        with pytest.raises(ValueError):
            rec.sweep_sharded(a_h[lo:hi], b_h[lo:hi], problem, out=(gb3, ga3))
        with pytest.raises(TypeError):
            rec.sweep_sharded(a_h[lo:hi], b_h[lo:hi], p.theta)
        rec.close()
        return e

    errs, fails, _ = _virtual_ranks(world, body)
    assert not any(fails), fails
    assert all(e is not None and e < TOL for e in errs), errs


@pytest.mark.gpu
@pytest.mark.parametrize("fail_rank,fail_call", [(1, 3), (0, 9), (2, 30)])
def test_native_sharded_sweep_collective_failure_reaches_every_rank(fail_rank, fail_call):
    """One rank's collective fails (its table entry returns an error, as a dead peer or a driver fault would): that rank aborts
    the communicator and returns VGPA_ERR_COMM; no rank stays inside a collective -- every rank comes back with an error, and the
    shard refuses further work.  (On real hardware the peers leave through ncclCommAbort + the shard's bounded waits.)"""
    from vgpa_amd.large_d import NativeShardedRecursion
    from test_gpu_edge_cases import make_problem
    d, n, world = 96, 7, 3
    p, x = make_problem("L96", d, n, method="rk4")
    comm = _CallbackComm(world, fail_at=(fail_rank, fail_call))

    def body(rank, comm):
        rec = NativeShardedRecursion("rk4", p.dt, d, n, rank=rank, world=world, device=0, comm=comm.table(rank))
        try:
            rec.sweep(x, p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), 0.0)
        except RuntimeError as exc:
            first = str(exc)
            with pytest.raises(RuntimeError):           # the shard is unusable from now on
                rec.sweep(x, p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), 0.0)
            return first
        finally:
            rec._h and rec._lib.vgpa_shard_destroy(rec._h)
            rec._h = None
        return None

    out, fails, _ = _virtual_ranks(world, body, comm)
    assert not any(fails), fails
    assert all(isinstance(o, str) for o in out), out          # every rank got an error, none hung, none "succeeded"
    assert "collective" in out[fail_rank]


@pytest.mark.gpu
def test_native_sharded_sweep_outcome_is_collective():
    """A covariance that loses positive definiteness on ONE rank's time slice: every rank raises LinAlgError -- the
    reference raises it and the whole run ends (variational.py:380) -- instead of one rank leaving and the others carrying on with
    NaN.  Then: unsorted / duplicate observation indices are refused before any collective (ValueError on every rank)."""
    from vgpa_amd.large_d import NativeShardedRecursion
    from test_gpu_edge_cases import make_problem
    d, n, world = 96, 9, 3
    p, x = make_problem("L96", d, n, method="euler")
    xb = x.copy()
    a = xb[:n * d * d].reshape(n, d, d)
    a[3:] = 150.0 * np.eye(d)            # Euler: S <- (1 - 2 * 150 * dt) S + ... = -2 S from grid point 4 on: rank 0's slice [0, 3) stays fine

    def body(rank, comm):
        rec = NativeShardedRecursion("euler", p.dt, d, n, rank=rank, world=world, device=0, comm=comm.table(rank))
        lo, hi = rec.time_slice
        kinds = []
        for xx, obs_t in ((xb, p.obs_t), (x, p.obs_t[::-1].copy()), (x, np.array([2, 2], dtype=np.int64))):
            try:
                rec.sweep(xx, p.theta, np.diag(p.sigma), p.m0, p.s0, obs_t, p.obs_y[:len(obs_t)], np.diag(p.obs_noise), 0.0)
                kinds.append("ok")
            except np.linalg.LinAlgError:
                kinds.append("notpd")
            except ValueError:
                kinds.append("arg")
        f, _, _ = rec.sweep(x, p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), 0.0)   # still usable
        rec.close()
        return (lo, hi), kinds, f

    out, fails, _ = _virtual_ranks(world, body)
    assert not any(fails), fails
    assert out[0][0] == (0, 3)
    assert all(o[1] == ["notpd", "arg", "arg"] for o in out), out
    f_o, _, _ = vo.sweep(p, x, faithful=False)
    e0 = float(np.asarray(vo.kl0(p)))
    assert all(abs(o[2] + e0 - f_o) <= TOL * abs(f_o) for o in out)


def _spawn_ranks(world, d, n, method, out_dir, fail=False, timeout=300):
    """`world` worker PROCESSES (tests/_shard_worker.py) on GPU 0; returns their exit codes."""
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), str(r), str(world), str(port), str(d),
                               str(n), method, str(out_dir)] + (["fail"] if fail else []), env=env) for r in range(world)]
    codes = []
    for pr in procs:
        try:
            codes.append(pr.wait(timeout=timeout))
        except subprocess.TimeoutExpired:
            pr.kill()
            codes.append(-9)
    return codes


@pytest.mark.gpu
@pytest.mark.parametrize("method,d,n,world", [("rk4", 128, 9, 2), ("heun", 192, 7, 3)])
def test_native_sharded_sweep_with_real_processes(method, d, n, world, tmp_path):
    """The native driver as it is deployed -- ONE PROCESS PER RANK -- on the one GPU of the test box: `vgpa_shard_sweep_sharded` in
    2 / 3 separate processes whose vgpa_comm table is HostStagedComm (gloo, data staged through the host; several ranks on one
    GPU cannot form an RCCL communicator).  Pipelined schedule (point-to-point groups -> batch_isend_irecv), exchanges, the
    agreement step: F on every rank and every rank's gradient slice vs the oracle."""
    from test_gpu_edge_cases import make_problem
    p, x = make_problem("L96", d, n, method=method)
    f_o, g_o, _ = vo.sweep(p, x, faithful=False)
    ga_o, gb_o = g_o[:n * d * d].reshape(n, d, d), g_o[n * d * d:].reshape(n, d)
    codes = _spawn_ranks(world, d, n, method, tmp_path)
    assert codes == [0] * world, codes
    covered = []
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert "error" not in z.files, str(z["error"]) if "error" in z.files else ""
        lo, hi = int(z["lo"]), int(z["hi"])
        covered.extend(range(lo, hi))
        assert int(z["chunks"]) > 0                                    # the pipelined gather ran
        assert abs(float(z["f"]) - f_o) <= TOL * abs(f_o)
        if hi > lo:
            assert rel_err(z["ga"], ga_o[lo:hi]) < TOL and rel_err(z["gb"], gb_o[lo:hi]) < TOL
    assert covered == list(range(n))


@pytest.mark.gpu
def test_native_sharded_sweep_a_dead_process_is_a_timeout_not_a_hang(tmp_path):
    """One of three rank processes exits before the sweep.  The survivors' collectives cannot complete; their bounded waits
    (VGPA_SHARD_OPT_TIMEOUT_MS = 20 s here) and the failing gloo calls turn that into VGPA_ERR_COMM -> RuntimeError on every
    surviving rank, and the processes end by themselves."""
    import time
    t0 = time.perf_counter()
    codes = _spawn_ranks(3, 96, 7, "rk4", tmp_path, fail=True, timeout=240)
    assert time.perf_counter() - t0 < 200
    assert codes[2] == 3 and codes[0] == 0 and codes[1] == 0, codes
    for r in (0, 1):
        z = np.load(tmp_path / f"r{r}.npz")
        assert "error" in z.files and "collective" in str(z["error"])


@pytest.mark.gpu
def test_native_sharded_fused_sweep_at_config5_matrix_size():
    """BASELINE configs[4] matrix size (D = 4096, RK4, 4-point grid): the fused sweep on 1, 2 and 8 virtual ranks against
    ORACLE anchors generated in the build container (tools/gen_d4096_anchor.py -> tests/golden/anchors_d4096.json: F, the
    Frobenius norm / max-abs / sampled entries of every grid point's gradient) -- blocked Cholesky, triangular inverse, SYRK
    and lde_grad at this size included -- and, on top, the sharded runs against the one-rank run at 1e-12 / 1e-11 (the sharding,
    the gathered / exchanged layouts and the sparse jumps change nothing beyond rounding)."""
    import json
    import os
    import sys
    import threading
    import torch
    from vgpa_amd.large_d import NativeShardedRecursion
    from conftest import GOLDEN_DIR, ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_d4096_anchor import inputs, D as d, N_PTS as n, METHOD as method
    anc = json.load(open(os.path.join(GOLDEN_DIR, "anchors_d4096.json")))
    assert anc["D"] == d and anc["Np"] == n
    xh, m0, s0, sig, obs_t, obs_y, rdiag = inputs()
    x = torch.as_tensor(xh, device="cuda")
    del xh
    ii = [i for i, _ in anc["samples_ij"]]
    jj = [j for _, j in anc["samples_ij"]]

    def sweep(world):
        comm = _CallbackComm(world)
        out, fails = [None] * world, []

        def run(rank):
            try:
                torch.cuda.set_device(0)
                rec = NativeShardedRecursion(method, 0.01, d, n, rank=rank, world=world, device=0,
                                             comm=comm.table(rank) if world > 1 else None)
                f, ga, gb = rec.sweep(x, 8.0, sig, m0, s0, obs_t, obs_y, rdiag, 0.0)
                out[rank] = (rec.time_slice, f, ga.cpu().numpy(), gb.cpu().numpy())
                rec.close()
            except BaseException as exc:      # noqa: BLE001
                fails.append(exc)
                comm.barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not fails, fails
        return out

    def against_oracle(lo, hi, f, ga, gb):
        assert abs(f - anc["F_minus_E0"]) <= TOL * abs(anc["F_minus_E0"]), (f, anc["F_minus_E0"])
        for t in range(lo, hi):
            a_t, b_t = ga[t - lo], gb[t - lo]
            assert abs(np.linalg.norm(a_t) - anc["grad_a_fro"][t]) <= TOL * anc["grad_a_fro"][t], t
            assert abs(np.abs(a_t).max() - anc["grad_a_absmax"][t]) <= TOL * anc["grad_a_absmax"][t], t
            assert abs(np.linalg.norm(b_t) - anc["grad_b_norm"][t]) <= TOL * anc["grad_b_norm"][t], t
            assert np.max(np.abs(a_t[ii, jj] - np.array(anc["grad_a_samples"][t]))) <= TOL * anc["grad_a_absmax"][t], t
            assert np.max(np.abs(b_t[:8] - np.array(anc["grad_b_first8"][t]))) <= TOL * anc["grad_b_absmax"][t], t

    (_, f1, ga1, gb1), = sweep(1)
    against_oracle(0, n, f1, ga1, gb1)
    for world in (2, 8):
        for (lo, hi), f, ga, gb in sweep(world):
            against_oracle(lo, hi, f, ga, gb)
            assert abs(f - f1) <= 1e-12 * abs(f1)
            if hi > lo:
                assert rel_err(ga, ga1[lo:hi]) < 1e-11 and rel_err(gb, gb1[lo:hi]) < 1e-11


@pytest.mark.gpu
def test_rccl_table_single_rank():
    """librccl behind vgpa_comm: unique id through the C ABI, communicator of one rank on this GPU, a grouped in-place
    all-gather and an all-to-all through the table's function pointers."""
    import ctypes
    import torch
    from vgpa_amd._lib import load, VgpaComm
    lib = load()
    uid = (ctypes.c_char * 128)()
    assert lib.vgpa_rccl_unique_id(uid) == 0
    comm = VgpaComm()
    assert lib.vgpa_rccl_comm_create(ctypes.byref(comm), uid, 0, 1, 0) == 0
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(x)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert comm.group_begin(comm.user) == 0
    assert comm.all_gather(comm.user, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), 1000, stream) == 0
    assert comm.group_end(comm.user) == 0
    assert comm.all_to_all(comm.user, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 1000, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, x) and float(x[999]) == 999.0
    n = ctypes.c_int(-1)
    assert lib.vgpa_rccl_comm_count(ctypes.byref(comm), ctypes.byref(n)) == 0 and n.value == 1     # ncclCommCount: bench.py's rccl_ranks
    assert lib.vgpa_rccl_comm_count(ctypes.byref(VgpaComm()), ctypes.byref(n)) == -1               # not an RCCL table
    # the table holds TWO communicators (ncclCommSplit): the point-to-point groups of the pipelined gather (communication stream) do
    # not share one with the compute stream's collectives -- exercised here through send / recv to the rank itself inside a group
    assert lib.vgpa_rccl_comm_streams(ctypes.byref(comm), ctypes.byref(n)) == 0 and n.value == 2
    z = torch.zeros_like(x)
    side = torch.cuda.Stream()
    s2 = ctypes.c_void_p(side.cuda_stream)
    assert comm.group_begin(comm.user) == 0
    assert comm.send(comm.user, ctypes.c_void_p(x.data_ptr()), 1000, 0, s2) == 0
    assert comm.recv(comm.user, ctypes.c_void_p(z.data_ptr()), 1000, 0, s2) == 0
    assert comm.group_end(comm.user) == 0
    assert comm.all_to_all(comm.user, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 1000, stream) == 0      # (the other communicator, concurrently)
    torch.cuda.synchronize()
    assert torch.equal(z, x) and torch.equal(y, x)
    lib.vgpa_rccl_comm_destroy(ctypes.byref(comm))


@pytest.mark.gpu
@pytest.mark.parametrize("world,chunks", [(1, None), (2, None), (4, 0)])
def test_sweep_phase_times_and_stage_timing_leave_the_results_alone(world, chunks):
    """bench.py's config-5 block reports where a sharded sweep's time goes: vgpa_shard_phase_ms (six phases of the last sweep, HIP
    events) and vgpa_shard_time_stage (one recursion stage with / without its collectives -- it overwrites the workspace).  Both are
    instrumentation: the sweep after them must still equal the oracle's, and the phases must add up to no more than the sweep."""
    import time
    from vgpa_amd.large_d import NativeShardedRecursion
    from vgpa_amd._lib import SHARD_OPT_GATHER_CHUNKS
    from test_gpu_edge_cases import make_problem
    d, n, method = 128, 12, "rk4"
    p, x = make_problem("L96", d, n, method=method)
    f_o, g_o, _ = vo.sweep(p, x, faithful=False)
    ga_o = g_o[:n * d * d].reshape(n, d, d)
    e0 = float(np.asarray(vo.kl0(p)))

    def body(rank, comm):
        rec = NativeShardedRecursion(method, p.dt, d, n, rank=rank, world=world, device=0, comm=comm.table(rank) if world > 1 else None)
        if chunks is not None:
            rec.set_option(SHARD_OPT_GATHER_CHUNKS, chunks)
        with pytest.raises(RuntimeError):
            rec.phase_ms()                       # no sweep yet: VGPA_ERR_STATE
        args = (x, p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y, np.diag(p.obs_noise), e0)
        rec.sweep(*args)
        t0 = time.perf_counter()
        _, ga0, gb0 = rec.sweep(*args)
        wall_ms = 1e3 * (time.perf_counter() - t0)
        ph = rec.phase_ms()
        t_with, t_without = rec.time_stage(3, True), rec.time_stage(3, False)
        # the workspace the timing runs scribbled over is re-initialised by the sweep (here with a prepared problem, into the pair of the sweep above)
        f, ga, _ = rec.sweep(x, rec.prepare(*args[1:]), out=(ga0, gb0))
        assert ga.data_ptr() == ga0.data_ptr()
        lo, hi = rec.time_slice
        e = max(abs(f - f_o) / abs(f_o), rel_err(ga.cpu().numpy(), ga_o[lo:hi]) if hi > lo else 0.0)
        ranks = rec.rccl_ranks
        rec.close()
        return e, ph, wall_ms, t_with, t_without, ranks

    out, fails, _ = _virtual_ranks(world, body)
    assert not any(fails), fails
    for e, ph, wall_ms, t_with, t_without, ranks in out:
        assert e < TOL
        assert list(ph) == ["x_exchange", "forward_recursion", "energy_obs", "exchanges", "backward_recursion", "gradient"]
        assert all(v >= 0.0 for v in ph.values()) and ph["forward_recursion"] > 0.0 and ph["backward_recursion"] > 0.0
        assert sum(ph.values()) <= wall_ms * 1.05 + 0.5
        assert t_with > 0.0 and t_without > 0.0
        assert ranks == 0                        # an injected table is not an RCCL communicator
