"""
GPU suite (-m gpu): edge cases of the hot path through the C ABI, all against the numpy oracle (lean mode).
  * every block geometry of the MFMA steppers and of the wave-level energy kernel: D = 10 ... 64
  * dense (non-diagonal) system noise Sigma, dense observation noise R, non-identity observation operator H
  * shortest grids (Np = 2, 3), a single observation, observations at the first / last index
  * odd batches, the four-waves-per-problem variant of the MFMA steppers, very small dt
"""
import numpy as np
import pytest

import vgpa_amd as va
from vgpa_amd._lib import FLAG_FORCE_GENERIC, FLAG_KEEP_PSI, FLAG_SYM_UNITS
from conftest import rel_err
from oracle import vgpa_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-9


def spd(rng, d, scale=1.0, jitter=0.3):
    q = rng.standard_normal((d, d)) / np.sqrt(d)
    return scale * (np.eye(d) + jitter * (q + q.T) / 2.0 + jitter * q.dot(q.T))


def make_problem(model, d, n_pts, method="rk4", dt=0.01, seed=5, dense=False, obs_at=None, h_op=None):
    rng = np.random.default_rng(seed)
    single = model in ("OU", "DW")
    if single:
        theta, sigma = 1.0, 0.8
        m0, s0, mu0, tau0 = 0.3, 0.2, 1.0, 0.5
        r = 0.04
    else:
        theta = np.array([10.0, 28.0, 2.667]) if model == "L63" else 8.0
        sigma = spd(rng, d, 4.0) if dense else np.diag(3.0 + rng.random(d))
        m0 = (8.0 if model == "L96" else 1.0) + rng.standard_normal(d)
        s0 = spd(rng, d, 0.2, 0.1) if dense else 0.2 * np.eye(d)
        mu0, tau0 = np.ones(d), 0.5 * np.eye(d)
        r = spd(rng, d, 1.0, 0.2) if dense else np.eye(d)
    obs_t = np.array(obs_at if obs_at is not None else sorted(set(range(2, n_pts - 1, 5))), dtype=np.int64)
    obs_y = rng.standard_normal(obs_t.size) if single else 8.0 + rng.standard_normal((obs_t.size, d))
    p = vo.Problem(model=model, method=method, dt=dt, theta=theta, sigma=sigma, m0=m0, s0=s0, mu0=mu0, tau0=tau0,
                   obs_t=obs_t, obs_y=obs_y, obs_noise=r, n_pts=n_pts, dim_d=1 if single else d, obs_h=h_op)
    if single:
        a = 1.6 + 0.05 * rng.standard_normal(n_pts)
        b = rng.standard_normal(n_pts)
        x = np.concatenate((a, b))
    else:
        a = (8.0 if model == "L96" else 20.0) * np.eye(d) + 0.05 * rng.standard_normal((n_pts, d, d))
        b = 8.0 * m0 + rng.standard_normal((n_pts, d))
        x = np.concatenate((a.ravel(), b.ravel()))
    return p, x


def gpu_context(p, batch=1, flags=0):
    d = p.dim_d
    sig = np.array([[p.sigma]]) if p.single_dim else p.sigma
    return va.Context(p.model, p.method, d, p.n_pts, p.dt, sigma=sig, theta=np.atleast_1d(p.theta),
                      m0=np.atleast_1d(p.m0), s0=np.asarray(p.s0, dtype=float).reshape(d, d), obs_t=p.obs_t,
                      obs_y=p.obs_y, obs_noise=np.asarray(p.obs_noise, dtype=float).reshape(d, d),
                      obs_h=None if p.obs_h is None else np.asarray(p.obs_h, dtype=float).reshape(d, d),
                      e0=float(np.asarray(vo.kl0(p))), batch=batch, flags=flags)


def check(p, x, flags=0, tol=TOL):
    ctx = gpu_context(p, flags=flags)
    f, g = ctx.sweep(x)
    f_ref, g_ref, st = vo.sweep(p, x, faithful=False)
    assert abs(f - f_ref) <= tol * abs(f_ref), (f, f_ref)
    assert rel_err(g, g_ref) < tol
    for key in ("mt", "st", "lamt", "psit"):
        got = ctx.fetch(key)
        want = st[key]
        assert rel_err(np.asarray(got).reshape(np.shape(want)), want) < tol, key
    ctx.close()


@pytest.mark.parametrize("d", [10, 11, 13, 16, 20, 21, 24, 28, 32, 33, 36, 44, 45, 52, 64])
@pytest.mark.parametrize("method", ["rk4", "heun"])
def test_every_block_geometry(d, method):
    n = 14 if d <= 44 else 8
    p, x = make_problem("L96", d, n, method=method)
    check(p, x)


@pytest.mark.parametrize("d", [33, 36, 40])
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
def test_role_specialised_steppers_at_33_to_40(d, method, monkeypatch):
    """33 <= D <= 40 defaults to the symmetric-unit cover kernels at every batch size; VGPA_ODE_KERNEL=pe (read at vgpa_create) keeps the
    role-specialised family reachable, and tested, there."""
    monkeypatch.setenv("VGPA_ODE_KERNEL", "pe")
    p, x = make_problem("L96", d, 14, method=method)
    check(p, x)


@pytest.mark.parametrize("d", [45, 48, 57, 64])
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
def test_matrix_core_steppers_above_44(d, method):
    """44 < D <= 64: the symmetric-unit kernels (ode_sym_impl.h) are the only matrix-core steppers there."""
    p, x = make_problem("L96", d, 9, method=method)
    check(p, x)


@pytest.mark.parametrize("d", [5, 8, 9, 12, 17, 24, 31, 36, 40, 44])
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
def test_symmetric_unit_steppers_every_geometry(d, method):
    """The symmetric-unit kernels where the role-specialised ones are the default (VGPA_FLAG_SYM_UNITS), odd D included."""
    p, x = make_problem("L96", d, 12, method=method)
    check(p, x, flags=FLAG_SYM_UNITS)


@pytest.mark.parametrize("model,d", [("L96", 12), ("L96", 40), ("L96", 52)])
def test_symmetric_unit_steppers_dense_inputs_and_batches(model, d):
    rng = np.random.default_rng(3)
    h = np.eye(d) + 0.1 * rng.standard_normal((d, d))
    p, x = make_problem(model, d, 16, dense=True, h_op=h)          # dense Sigma / S0 / R / H: dense matrix jumps
    check(p, x, flags=FLAG_SYM_UNITS)
    p, x = make_problem(model, d, 23, method="rk4")
    ctx = gpu_context(p, batch=3, flags=FLAG_SYM_UNITS)
    xb = np.stack([x + 0.01 * rng.standard_normal(x.size) for _ in range(3)])
    fb, gb = ctx.sweep(xb)
    for i in range(3):
        f_ref, g_ref, _ = vo.sweep(p, xb[i], faithful=False)
        assert abs(fb[i] - f_ref) <= TOL * abs(f_ref)
        assert rel_err(gb[i], g_ref) < TOL
    ctx.close()


@pytest.mark.parametrize("d,method", [(40, "rk4"), (36, "rk2"), (33, "rk4")])
def test_q_stream_of_the_batched_sweeps(d, method):
    """33 <= D <= 40, RK2 / RK4 on the symmetric-unit kernels: the backward kernel leaves Q''_t = Sigma^-1 A_t - 2 Psi_t where Psi_t
    would be (the gradient assembly then reads one matrix stream less) and VGPA_FETCH_PSIT recovers Psi_t.  Against the same
    sweep with VGPA_FLAG_KEEP_PSI (Psi_t stored, A_t re-read): F identical, gradient and Psi_t equal to rounding; a second fetch
    and a gradient(eval_fun=False) after the recovery still see consistent data."""
    p, x = make_problem("L96", d, 21, method=method)
    ctx_q = gpu_context(p, flags=FLAG_SYM_UNITS)
    ctx_k = gpu_context(p, flags=FLAG_SYM_UNITS | FLAG_KEEP_PSI)
    f_q, g_q = ctx_q.sweep(x)
    f_k, g_k = ctx_k.sweep(x)
    assert f_q == f_k
    assert rel_err(g_q, g_k) < 1e-13
    psi_q, psi_k = ctx_q.fetch("psit"), ctx_k.fetch("psit")
    assert rel_err(psi_q, psi_k) < 1e-13
    assert np.array_equal(ctx_q.fetch("psit"), psi_q)                      # recovered once, in place
    # dEsde_dS: ctx_q's energy kernel writes packed lower triangles (unpacked on the way out), ctx_k's the upper triangle of whole
    # matrices (mirrored on the way out): the two orientations of the SYRK round the scaled operand differently -- equal to rounding
    ds_q, ds_k = ctx_q.fetch("dEsde_ds"), ctx_k.fetch("dEsde_ds")
    assert np.array_equal(ds_q, np.swapaxes(ds_q, 1, 2))
    assert rel_err(ds_q, ds_k) < 1e-13
    assert np.array_equal(ctx_q.fetch("dEsde_ds"), ds_q)                     # (the packed stream is left as it is: a second fetch sees the same)
    assert rel_err(ctx_q.gradient(None), g_k) < 1e-13                      # assembled from the recovered Psi_t now
    _, g_ref, st = vo.sweep(p, x, faithful=False)
    assert rel_err(g_q, g_ref) < TOL and rel_err(np.asarray(psi_q).reshape(np.shape(st["psit"])), st["psit"]) < TOL
    ctx_q.close(); ctx_k.close()


@pytest.mark.parametrize("d,n_pts,obs", [(40, 2, [1]), (40, 3, [0]), (40, 4, [1, 2]), (40, 5, [3]), (40, 24, None), (33, 3, [1]), (33, 11, None),
                                         (36, 4, [2]), (37, 9, None), (39, 2, [0])])
def test_gradient_waves_of_the_backward_kernel(d, n_pts, obs):
    """Batches of >= 64 Lorenz-96 problems with 33 <= D <= 40 under RK4: the backward kernel assembles the gradient on a third set
    of waves (k_ode_sym, GF; grad_waves) -- a two-step pipeline per grid point with its own first / second / last steps, so every
    short grid is a case of its own, and every D < 40 exercises the padding.  Every problem against the same sweep with
    VGPA_FLAG_KEEP_PSI (backward kernel + separate assembly), some against the oracle; lam_t / Psi_t, which the fused kernel keeps to
    itself, as vgpa_fetch materialises them; free_energy (no backward recursion at all) followed by gradient(None)."""
    batch = 67
    p, x = make_problem("L96", d, n_pts, method="rk4", obs_at=obs)
    rng = np.random.default_rng(23)
    xb = x[None, :] + 0.02 * rng.standard_normal((batch, x.size))
    ctx, ctx_k = gpu_context(p, batch=batch), gpu_context(p, batch=batch, flags=FLAG_KEEP_PSI)
    fb, gb = ctx.sweep(xb)
    fk, gk = ctx_k.sweep(xb)
    assert np.array_equal(fb, fk)
    assert max(rel_err(gb[i], gk[i]) for i in range(batch)) < 1e-12
    for i in (0, 31, batch - 1):
        f_ref, g_ref, st = vo.sweep(p, xb[i], faithful=False)
        assert abs(fb[i] - f_ref) <= TOL * abs(f_ref) and rel_err(gb[i], g_ref) < TOL
        if i == 0:
            lam, psi = ctx.fetch("lamt"), ctx.fetch("psit")
            assert rel_err(lam[0].reshape(np.shape(st["lamt"])), st["lamt"]) < TOL
            assert rel_err(psi[0].reshape(np.shape(st["psit"])), st["psit"]) < TOL
    assert np.array_equal(ctx.gradient(None), gb)            # (after the fetch: the fused kernel again, same bits)
    f2 = ctx.free_energy(xb[::-1].copy())
    g2 = ctx.gradient(None)
    assert np.array_equal(f2, fb[::-1]) and np.array_equal(g2, gb[::-1])
    ctx.close(); ctx_k.close()


def test_fused_gradient_kernel_below_its_default_batch_size():
    """Below 64 problems per context the default is the backward kernel followed by the separate assembly; VGPA_FUSED_GRAD=1 forces the
    kernel with the gradient waves (read once per process: a child process).  One problem, three problems and a padded dimension
    against the default path of this process and the oracle."""
    import json
    import os
    import subprocess
    import sys
    cases = ((40, 9, 1), (40, 14, 3), (35, 6, 2))
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_edge_cases as t\n"
        "out = []\n"
        "for d, n, batch in %r:\n"
        "    p, x = t.make_problem('L96', d, n, method='rk4')\n"
        "    ctx = t.gpu_context(p, batch=batch)\n"
        "    xb = np.stack([x + 0.01 * i for i in range(batch)]) if batch > 1 else x\n"
        "    f, g = ctx.sweep(xb)\n"
        "    out.append({'f': [float(v) for v in np.atleast_1d(f)], 'g': np.asarray(g).ravel().tolist()})\n"
        "    ctx.close()\n"
        "print(json.dumps(out))\n" % (os.path.dirname(__file__), cases))
    env = dict(os.environ)
    env["VGPA_FUSED_GRAD"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    fused = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("[")][-1])
    for (d, n, batch), got in zip(cases, fused):
        p, x = make_problem("L96", d, n, method="rk4")
        ctx = gpu_context(p, batch=batch)
        xb = np.stack([x + 0.01 * i for i in range(batch)]) if batch > 1 else x
        f, g = ctx.sweep(xb)
        ctx.close()
        assert np.array_equal(np.atleast_1d(f), np.asarray(got["f"]))                   # F: the same kernels either way
        assert rel_err(np.asarray(got["g"]), np.asarray(g).ravel()) < 1e-12              # the gradient: another product order
        f_ref, g_ref, _ = vo.sweep(p, np.atleast_2d(xb)[0], faithful=False)
        assert rel_err(np.asarray(got["g"]).reshape(batch, -1)[0], g_ref) < TOL


@pytest.mark.parametrize("method", ["rk4", "heun"])
def test_more_problems_than_compute_units(method):
    """bench.py's regime: a batch larger than twice the CU count, so that the default dispatch picks the symmetric-unit steppers and
    two (in the last round of the grid: one) workgroups share a CU.  EVERY problem's F and gradient against the oracle."""
    import torch
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    batch = 2 * n_cu + 37
    p, x = make_problem("L96", 40, 12, method=method)
    rng = np.random.default_rng(17)
    xb = x[None, :] + 0.02 * rng.standard_normal((batch, x.size))
    ctx = gpu_context(p, batch=batch)
    fb, gb = ctx.sweep(xb)
    worst = 0.0
    for i in range(batch):
        f_ref, g_ref, _ = vo.sweep(p, xb[i], faithful=False)
        worst = max(worst, abs(fb[i] - f_ref) / abs(f_ref), rel_err(gb[i], g_ref))
    assert worst < TOL, worst
    ctx.close()


@pytest.mark.parametrize("model,d", [("L96", 12), ("L96", 40), ("L63", 3)])
def test_dense_noise_matrices_and_observation_operator(model, d):
    rng = np.random.default_rng(11)
    h = np.eye(d) + 0.1 * rng.standard_normal((d, d))
    p, x = make_problem(model, d, 25, dense=True, h_op=h)
    check(p, x)                                   # dense Sigma, S0, R, H: symmetric inputs -> MFMA steppers
    check(p, x, flags=FLAG_FORCE_GENERIC)


@pytest.mark.parametrize("n_pts,obs", [(2, [0]), (3, [1]), (6, [0, 4]), (12, [10])])
@pytest.mark.parametrize("model,d", [("OU", 1), ("L63", 3), ("L96", 12)])
def test_shortest_grids_and_boundary_observations(model, d, n_pts, obs):
    p, x = make_problem(model, d, n_pts, obs_at=obs)
    check(p, x)


@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
def test_all_steppers_double_well_and_small_dt(method):
    p, x = make_problem("DW", 1, 40, method=method, dt=1e-4)
    check(p, x)


@pytest.mark.parametrize("method", ["rk4", "heun", "rk2", "euler"])
@pytest.mark.parametrize("batch", [1, 2, 5])
def test_odd_batches(batch, method):
    p, x = make_problem("L96", 12, 20, method=method)
    ctx = gpu_context(p, batch=batch)
    rng = np.random.default_rng(1)
    xb = np.stack([x + 0.01 * rng.standard_normal(x.size) for _ in range(batch)])
    fb, gb = ctx.sweep(xb if batch > 1 else xb[0])
    fb, gb = np.atleast_1d(fb), np.atleast_2d(gb)
    for i in range(batch):
        f_ref, g_ref, _ = vo.sweep(p, xb[i], faithful=False)
        assert abs(fb[i] - f_ref) <= TOL * abs(f_ref)
        assert rel_err(gb[i], g_ref) < TOL


def test_batched_operator_level_sweeps_with_dense_jumps():
    """FwdOde / BwdOde semantics (dense jump arrays) through the MFMA stepping kernels, two problems in one context."""
    from test_large_d import make_inputs
    d, n = 24, 11
    a, b, m0, s0, sigma, gm, gs, jm, js = make_inputs(d, n)
    ctx = va.Context("NONE", "rk4", d, n, 0.01, sigma=np.eye(d), batch=2)
    a2 = np.stack([a, a * 1.01])
    lam, psi = ctx.solve_bwd(a2, np.stack([gm, gm]), np.stack([gs, gs]), np.stack([jm, jm]), np.stack([js, js]))
    for i in range(2):
        lam_o, psi_o = vo.solve_bwd("rk4", 0.01, False, a2[i], gm, gs, jm, js)
        assert rel_err(lam[i], lam_o) < TOL and rel_err(psi[i], psi_o) < TOL
    mt, st = ctx.solve_fwd(a2, np.stack([b, b]), m0, s0, sigma)
    for i in range(2):
        mt_o, st_o = vo.solve_fwd("rk4", 0.01, False, a2[i], b, m0, s0, sigma)
        assert rel_err(mt[i], mt_o) < TOL and rel_err(st[i], st_o) < TOL


def test_state_errors():
    p, x = make_problem("L63", 3, 10)
    ctx = gpu_context(p)
    with pytest.raises(RuntimeError):
        ctx.gradient(None)                         # gradient(x, eval_fun=False) before any free_energy
    with pytest.raises(ValueError):
        ctx.free_energy(x[:-1])
    with pytest.raises(ValueError):
        va.Context("L63", "rk4", 3, 10, 0.01, sigma=np.eye(3), theta=[1.0])     # L63 needs 3 drift parameters
    with pytest.raises(ValueError):
        va.Context("L96", "rk4", 12, 10, -0.01, sigma=np.eye(12), theta=[8.0])
    p72, x72 = make_problem("L96", 72, 5, dense=True)
    c72 = gpu_context(p72)
    c72.free_energy(x72)
    g72 = c72.gradient(None)                       # D > 64 with a dense system noise matrix: built since round 3
    assert rel_err(g72, vo.sweep(p72, x72, faithful=False)[1]) < TOL
    with pytest.raises(NotImplementedError):       # what the large-D path still refuses: a batch in the time-chunked sweep
        from vgpa_amd._lib import FLAG_STREAM_LARGE_D
        gpu_context(make_problem("L96", 72, 5)[0], batch=2, flags=FLAG_STREAM_LARGE_D)


@pytest.mark.parametrize("d", [2, 3, 4])
@pytest.mark.parametrize("method", ["euler", "heun", "rk2", "rk4"])
@pytest.mark.parametrize("nb", [6, 520])
def test_lane_per_problem_steppers(d, method, nb):
    """D <= 4: >= 512 problems run one LANE per problem (ode_small.hip), fewer run 16 lanes per problem with shuffled
    operands (ode_wave.hip; nb = 6 leaves half of the second wave idle).  Operator-level calls with non-symmetric inputs
    and dense jumps: a few problems against the oracle, all of them against the workgroup-per-problem kernels
    (VGPA_FLAG_FORCE_GENERIC), which evaluate the same expressions in the same order."""
    rng = np.random.default_rng(100 * d + len(method))
    n = 24
    a = 2.0 * np.eye(d) + 0.3 * rng.standard_normal((nb, n, d, d))
    b = rng.standard_normal((nb, n, d))
    m0 = rng.standard_normal(d)
    s0 = 0.2 * np.eye(d) + 0.01 * rng.standard_normal((d, d))
    sigma = np.eye(d) + 0.1 * rng.standard_normal((d, d))
    gm = rng.standard_normal((nb, n, d))
    gs = rng.standard_normal((nb, n, d, d))
    jm = np.zeros((nb, n, d)); jm[:, 7] = rng.standard_normal((nb, d))
    js = np.zeros((nb, n, d, d)); js[:, 7] = rng.standard_normal((nb, d, d)); js[:, 0] = rng.standard_normal((nb, d, d))
    res = []
    for flags in (0, FLAG_FORCE_GENERIC):
        ctx = va.Context("NONE", method, d, n, 0.01, sigma=sigma, batch=nb, flags=flags)
        mt, st = ctx.solve_fwd(a, b, m0, s0, sigma)
        lam, psi = ctx.solve_bwd(a, gm, gs, jm, js)
        ctx.close()
        res.append((mt, st, lam, psi))
    for got, want in zip(res[0], res[1]):
        assert rel_err(got, want) < 1e-13
    for p in sorted({0, min(63, nb - 1), min(64, nb - 1), nb - 1}):
        mt_o, st_o = vo.solve_fwd(method, 0.01, False, a[p], b[p], m0, s0, sigma)
        lam_o, psi_o = vo.solve_bwd(method, 0.01, False, a[p], gm[p], gs[p], jm[p], js[p])
        assert rel_err(res[0][0][p], mt_o) < TOL and rel_err(res[0][1][p], st_o) < TOL
        assert rel_err(res[0][2][p], lam_o) < TOL and rel_err(res[0][3][p], psi_o) < TOL


def test_lane_per_problem_sweep_of_lorenz63():
    """The fused sweep of 600 Lorenz-63 problems (lane-per-problem steppers, sparse jumps) equals the single-problem
    contexts (workgroup-per-problem steppers) for a few of them."""
    from helpers import build_problem
    p = build_problem("L63", "RK4", 0.6, 0.01, None)
    v = p["vgp"]
    x0 = v.initialization()
    nb = 600
    xb = np.stack([x0 + 0.05 * np.random.default_rng(i).standard_normal(x0.size) for i in range(nb)])
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    kw = dict(sigma=p["model"].sigma, theta=p["model"].theta, m0=p["m0"], s0=p["s0"], obs_t=p["obs_t"], obs_y=p["obs_y"],
              obs_noise=p["obs_noise"], e0=e0)
    ctx = va.Context("L63", "rk4", 3, v.dim_n, 0.01, batch=nb, **kw)
    f, g = ctx.sweep(xb)
    ctx.close()
    one = va.Context("L63", "rk4", 3, v.dim_n, 0.01, batch=1, **kw)
    for i in (0, 1, 64, 599):
        f1, g1 = one.sweep(xb[i])
        assert abs(f[i] - f1) <= 1e-12 * abs(f1) and rel_err(g[i], g1) < 1e-12
    one.close()


def test_lane_pass_with_a_non_symmetric_initial_covariance():
    """The fused lane pass keeps S_t as its lower triangle: a Lorenz-63 context with a NON-symmetric s0 (the reference takes any matrix:
    ode_solver.py:60 evaluates both products of the slope literally) must not be symmetrised behind the caller's back -- it takes
    the four-kernel path, exactly as a VGPA_FLAG_MATERIALIZE context does, and agrees with the oracle."""
    from vgpa_amd._lib import FLAG_MATERIALIZE
    p, x = make_problem("L63", 3, 23, method="rk4")
    s0 = np.array(p.s0, dtype=float)
    s0[0, 1] += 0.03; s0[2, 0] -= 0.02                     # not symmetric
    p.s0 = s0
    nb = 520
    xb = x[None, :] + 0.02 * np.random.default_rng(3).standard_normal((nb, x.size))
    ctx, ref = gpu_context(p, batch=nb), gpu_context(p, batch=nb, flags=FLAG_MATERIALIZE)
    f, g = ctx.sweep(xb)
    fr, gr = ref.sweep(xb)
    assert np.array_equal(f, fr) and np.array_equal(g, gr)
    st = ctx.fetch("st")
    assert np.max(np.abs(st[0] - np.swapaxes(st[0], 1, 2))) > 1e-3          # S_t really is not symmetric
    f_ref, g_ref, _ = vo.sweep(p, xb[7], faithful=False)
    assert abs(f[7] - f_ref) <= TOL * abs(f_ref) and rel_err(g[7], g_ref) < TOL
    ctx.close(); ref.close()


def test_diagnostic_phase_repeat_leaves_the_results_alone():
    """VGPA_DIAG_REPEAT=<phase>:<n> (tools/power_per_kernel.sh: one kernel of the fused sweep held on the chip for clock / power
    samples) launches a phase n times; every phase is a pure function of its inputs, so F and the gradient must not move in any
    bit.  The variable is read once per process: child processes."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_edge_cases as t\n"
        "p, x = t.make_problem('L96', 40, 30)\n"
        "ctx = t.gpu_context(p, batch=2)\n"
        "f, g = ctx.sweep(np.stack([x, x + 0.01]))\n"
        "print(json.dumps({'f': [float(v).hex() for v in f], 'g': float(np.abs(g).sum()).hex()}))\n" % os.path.dirname(__file__))
    outs = []
    for spec in ("", "fwd:3", "energy:2", "bwd:3", "grad:2"):
        env = dict(os.environ)
        env.pop("VGPA_DIAG_REPEAT", None)
        if spec:
            env["VGPA_DIAG_REPEAT"] = spec
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert all(o == outs[0] for o in outs[1:]), outs


def test_stepper_variants_of_the_fragment_cover_agree():
    """33 <= D <= 40 on the fragment-cover steppers: with four or eight helper waves beside the four product waves of a workgroup (the
    default up to one problem per CU: the chores of a stage off the product waves' issue slots) and without them the same operations run in the same
    order -- F and the gradient must not differ in any bit.  (The outer-product cover and the eight-product-wave split, measured and
    rejected, are no longer part of the product build: -DVGPA_EXPERIMENTS.)  The switch is read once per process: child processes;
    RK4 and Heun, an unpadded and a padded dimension, one problem and a small batch."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_edge_cases as t\n"
        "out = {}\n"
        "for d, method, batch in ((40, 'rk4', 1), (37, 'heun', 3), (33, 'rk2', 2)):\n"
        "    p, x = t.make_problem('L96', d, 14, method=method)\n"
        "    ctx = t.gpu_context(p, batch=batch)\n"
        "    xb = np.stack([x + 0.01 * i for i in range(batch)]) if batch > 1 else x\n"
        "    f, g = ctx.sweep(xb)\n"
        "    out['%%d %%s' %% (d, method)] = {'f': [float(v) for v in np.atleast_1d(f)], 'g': np.asarray(g).ravel().tolist()}\n"
        "    ctx.close()\n"
        "print(json.dumps(out))\n" % os.path.dirname(__file__))
    outs = {}
    for name, env_set in (("helpers", {"VGPA_SYM_HELPERS": "1"}), ("two", {"VGPA_SYM_HELPERS": "2"}), ("plain", {"VGPA_SYM_HELPERS": "0"})):
        env = dict(os.environ)
        for k in ("VGPA_SYM_HELPERS", "VGPA_SYM_COVER", "VGPA_SYM_WAVES"):
            env.pop(k, None)
        env.update(env_set)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for key, ref in outs["plain"].items():
        assert outs["helpers"][key] == ref, key                      # bit for bit
        assert outs["two"][key] == ref, key                          # (two helper roles: vector recursion and staging on waves of their own)


@pytest.mark.parametrize("model,method,n,nb", [("L63", "rk4", 37, 520), ("L63", "rk4", 6, 576), ("L63", "heun", 22, 513), ("L63", "rk2", 9, 640),
                                               ("L63", "euler", 3, 512), ("OU", "euler", 41, 70), ("OU", "rk4", 18, 1), ("DW", "rk4", 33, 130),
                                               ("DW", "heun", 17, 64), ("OU", "rk2", 3, 3)])
def test_fused_lane_pass(model, method, n, nb):
    """BASELINE configs[0] / [1] batched: OU, double well and Lorenz-63 contexts on the lane-per-problem path run the objective as
    forward kernel -> observations -> ONE fused pass (E_sde terms in registers, backward recursion, gradient, F); the streams
    travel through LDS in chunks of T grid points.  Grids shorter than, equal to and not a multiple of the chunk; batches that
    leave the last wave ragged.  Checked: a few problems against the ORACLE, every problem against the four-kernel path
    (VGPA_FLAG_MATERIALIZE: same expressions, F summed in another order), free_energy + gradient(eval_fun=False) against the
    one-call sweep, and the arrays the fused pass never wrote (lam_t, Psi_t, dEsde_dm, dEsde_dS, <f>, E_sde(t)) as vgpa_fetch
    materialises them."""
    from vgpa_amd._lib import FLAG_MATERIALIZE
    d = 3 if model == "L63" else 1
    p, x0 = make_problem(model, d, n, method=method, obs_at=sorted(set(range(1, n - 1, 4))) if n > 3 else [1])
    rng = np.random.default_rng(n * 1000 + nb)
    xb = x0[None, :] + 0.05 * rng.standard_normal((nb, x0.size))
    ctx, ref = gpu_context(p, batch=nb), gpu_context(p, batch=nb, flags=FLAG_MATERIALIZE)
    f, g = ctx.sweep(xb)
    f_r, g_r = ref.sweep(xb)
    f, g, f_r, g_r = np.atleast_1d(f), np.atleast_2d(g), np.atleast_1d(f_r), np.atleast_2d(g_r)
    assert np.max(np.abs(f - f_r) / np.abs(f_r)) < 1e-11
    assert rel_err(g, g_r) < 1e-11
    for i in sorted({0, min(63, nb - 1), min(64, nb - 1), nb - 1}):
        f_o, g_o, st = vo.sweep(p, xb[i], faithful=False)
        assert abs(f[i] - f_o) <= TOL * abs(f_o), (i, f[i], f_o)
        assert rel_err(g[i], g_o) < TOL, i
    # two calls: F alone (no recursion), then the gradient from the cached moments
    f2 = np.atleast_1d(ctx.free_energy(xb))
    g2 = np.atleast_2d(ctx.gradient())
    assert rel_err(f2, f) < 1e-13 and np.array_equal(g2, g)      # (two instantiations of the pass: F to rounding, the gradient's kernel is the same)
    # what the fused pass keeps in registers, on demand
    for key in ("lamt", "psit", "dEsde_dm", "dEsde_ds", "Efx", "Esde_t", "mt", "st", "Edf"):
        got, want = ctx.fetch(key), ref.fetch(key)
        assert rel_err(got, want) < 1e-11, key
    e0, es, eo = ctx.energy_parts()
    e0r, esr, eor = ref.energy_parts()
    assert rel_err(np.atleast_1d(es), np.atleast_1d(esr)) < 1e-11 and rel_err(np.atleast_1d(eo), np.atleast_1d(eor)) < 1e-12
    # and a gradient behind the fetch (derived arrays valid: either route) still equals the sweep's
    assert rel_err(np.atleast_2d(ctx.gradient()), g) < 1e-11
    ctx.close(); ref.close()
