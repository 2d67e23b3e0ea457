"""the SCG iteration with device-resident vectors must follow the host SCG trace."""
import numpy as np
import pytest

import vgpa_amd as va
from helpers import build_problem

pytestmark = pytest.mark.gpu
OPTS = {"max_it": 12, "x_tol": 1e-6, "f_tol": 1e-8}


def _segments(ctx, host):
    from vgpa_amd._lib import DeviceBuffer
    buf = DeviceBuffer(ctx, host.size)
    buf.upload(host)
    return buf


def test_vector_algebra_matches_numpy():
    p = build_problem("OU", "Euler", 1.0, 0.01, None)
    v = va.VarGP(p["model"], p["m0"], p["s0"], p["fwd"], p["bwd"], p["lik"], p["kl0"], p["obs_y"], p["obs_t"], batch=3)
    ctx = v._context()
    rng = np.random.default_rng(5)
    for n in (1, 7, 255, 256, 257, 100003):
        a, b = rng.standard_normal((3, n)), rng.standard_normal((3, n))
        da, db, dc = _segments(ctx, a), _segments(ctx, b), _segments(ctx, np.zeros((3, n)))
        assert np.allclose(ctx.vdot(da, db), (a * b).sum(1), rtol=1e-12, atol=1e-12 * n)
        assert np.array_equal(ctx.vabsmax(da), np.abs(a).max(1))
        assert np.allclose(ctx.vasum(da), np.abs(a).sum(1), rtol=1e-13)
        al, be = np.array([0.5, 0.0, -2.0]), np.array([1.0, 1.0, 0.0])
        ctx.vaxpby(al, da, be, db, dc)
        want = al[:, None] * a + be[:, None] * b
        got = dc.download().reshape(3, n)
        assert np.array_equal(got[1], b[1]) and np.array_equal(got[2], -2.0 * a[2])
        assert np.allclose(got, want, rtol=1e-15, atol=0)
        ctx.vaxpby(-1.0, da, None, None, da)            # in place, scalar coefficient, no second operand
        assert np.array_equal(da.download().reshape(3, n), -a)
        for buf in (da, db, dc):
            buf.free()
    with pytest.raises(ValueError):
        ctx.vdot(_segments(ctx, np.zeros(4)), _segments(ctx, np.zeros(4)))


@pytest.mark.parametrize("name,method,tf,d", [("OU", "Euler", 2.0, None), ("DW", "Heun", 1.0, None),
                                              ("L63", "RK4", 0.5, None), ("L96", "RK2", 0.3, 12)])
def test_device_scg_follows_the_host_trace(name, method, tf, d):
    p = build_problem(name, method, tf, 0.01, d)
    v = p["vgp"]
    x0 = v.initialization()
    host = va.SCG(v.free_energy, v.gradient, dict(OPTS))
    x_h, f_h = host(x0.copy())
    dev = v.device_scg(dict(OPTS))
    x_d, f_d = dev(x0.copy())
    n_it = int(dev.statistics["MaxIt"][0])
    assert n_it == host.statistics["MaxIt"]
    assert np.allclose(dev.statistics["fx"][:n_it, 0], host.statistics["fx"][:n_it], rtol=1e-9, atol=0)
    assert np.allclose(dev.statistics["beta"][:n_it, 0], host.statistics["beta"][:n_it], rtol=1e-9)
    assert np.allclose(dev.statistics["dfx"][:n_it, 0], host.statistics["dfx"][:n_it], rtol=1e-7)
    assert abs(f_d - f_h) <= 1e-9 * abs(f_h)
    assert np.abs(x_d - x_h).max() <= 1e-7 * np.abs(x_h).max()
    assert f_d <= v.free_energy(x0)
    # one forward-backward evaluation less per accepted step than the reference's call pattern
    assert dev.statistics["f_eval"] <= host.statistics["f_eval"] - np.count_nonzero(np.diff(host.statistics["fx"][:n_it]))
    # the state left behind belongs to an evaluated point: arg_out can be fetched
    assert np.isfinite(v.arg_out["mt"]).all()


def test_batched_device_scg_equals_independent_runs():
    """B problems in lock step (different starting points, different convergence histories) == B single runs."""
    p = build_problem("L63", "RK4", 0.4, 0.01, None)
    args = (p["model"], p["m0"], p["s0"], p["fwd"], p["bwd"], p["lik"], p["kl0"], p["obs_y"], p["obs_t"])
    v1 = va.VarGP(*args)
    x0 = v1.initialization()
    rng = np.random.default_rng(9)
    starts = np.stack([x0, x0 + 0.02 * rng.standard_normal(x0.size), x0 + 0.2 * rng.standard_normal(x0.size)])
    opts = {"max_it": 10, "x_tol": 1e-3, "f_tol": 1e-2}        # loose: the problems stop at different iterations
    singles = []
    for s in starts:
        run = v1.device_scg(dict(opts))
        singles.append((run(s.copy()), int(run.statistics["MaxIt"][0])))
    vb = va.VarGP(*args, batch=3)
    run = vb.device_scg(dict(opts))
    xb, fb = run(starts.copy())
    for k, ((xs, fs), its) in enumerate(singles):
        assert int(run.statistics["MaxIt"][k]) == its
        assert abs(fb[k] - fs) <= 1e-9 * abs(fs)
        assert np.abs(xb[k] - xs).max() <= 1e-7 * np.abs(xs).max()


@pytest.mark.parametrize("stream", [False, True])
def test_device_scg_on_the_large_d_path(stream):
    """D > 64: the GEMM-per-stage drivers (resident and time-chunked) behind the same optimiser."""
    from vgpa_amd._lib import FLAG_STREAM_LARGE_D
    p = build_problem("L96", "RK4", 0.5, 0.01, 72)
    args = (p["model"], p["m0"], p["s0"], p["fwd"], p["bwd"], p["lik"], p["kl0"], p["obs_y"], p["obs_t"])
    v = va.VarGP(*args, flags=FLAG_STREAM_LARGE_D if stream else 0)
    assert v._context().streaming == stream
    x0 = v.initialization()
    opts = {"max_it": 4, "x_tol": 1e-6, "f_tol": 1e-8}
    host = va.SCG(v.free_energy, v.gradient, dict(opts))
    x_h, f_h = host(x0.copy())
    dev = v.device_scg(dict(opts))
    x_d, f_d = dev(x0.copy())
    assert abs(f_d - f_h) <= 1e-9 * abs(f_h) and np.abs(x_d - x_h).max() <= 1e-7 * np.abs(x_h).max()
    assert f_d < v.free_energy(x0)


def test_optimiser_trace_equals_the_references_at_baseline_size():
    """Three SCG iterations at BASELINE configs[2] (L96, D=40, RK4, Np=1001): the objective after every iteration, the
    trust-region scale and the final x must equal what the REFERENCE produced (tests/golden/scg_trace_config3.json,
    generated by tools/gen_scg_anchor.py in 101 s of CPU time) -- for the host SCG and for the device-resident one."""
    import json
    import os
    from conftest import GOLDEN_DIR
    ref = json.load(open(os.path.join(GOLDEN_DIR, "scg_trace_config3.json")))
    p = build_problem("L96", "RK4", 10.0, 0.01, 40)
    v = p["vgp"]
    x0 = v.initialization()
    opts = {"max_it": 3, "x_tol": 1e-6, "f_tol": 1e-8, "display": False}
    for make in (lambda: va.SCG(v.free_energy, v.gradient, dict(opts)), lambda: v.device_scg(dict(opts))):
        opt = make()
        x, fx = opt(x0.copy())
        st = opt.statistics
        assert np.allclose(np.asarray(st["fx"])[:3].ravel(), ref["fx_trace"], rtol=1e-9, atol=0)
        assert np.allclose(np.asarray(st["beta"])[:3].ravel(), ref["beta_trace"], rtol=1e-12, atol=0)
        assert abs(fx - ref["f_final"]) <= 1e-9 * abs(ref["f_final"])
        assert abs(np.linalg.norm(x) - ref["x_norm"]) <= 1e-11 * ref["x_norm"]


def test_whole_optimisation_equals_the_references_at_baseline_size():
    """The reference's complete optimisation at BASELINE configs[2] (SCG until its own termination: 30 iterations, 50
    objective evaluations, 523 s of CPU time; tests/golden/scg_full_config3.json): same objective after every iteration,
    same trust-region scale, same minimum; the iteration count may differ by the length of the stalled tail."""
    import json
    import os
    from conftest import GOLDEN_DIR
    ref = json.load(open(os.path.join(GOLDEN_DIR, "scg_full_config3.json")))
    n_it = ref["MaxIt_stat"]
    p = build_problem("L96", "RK4", 10.0, 0.01, 40)
    v = p["vgp"]
    x0 = v.initialization()
    opts = {"max_it": 500, "x_tol": 1e-6, "f_tol": 1e-8, "display": False}
    host = va.SCG(v.free_energy, v.gradient, dict(opts))
    x, fx = host(x0.copy())
    st = host.statistics
    # The last ~20 iterations make no progress: every step is rejected (F(x + step) > F(x) by rounding noise), beta grows by
    # exactly 4 per rejected step, and the run ends at the first step whose F is not above the old one -- at the latest when
    # the step (proportional to 1 / beta) falls under half an ulp of x, ~27 rejections behind the stall.  WHEN that first
    # happens is decided by the last bits of F ~ 3.7e4 (the kernels sum in a different order than numpy; the reference itself
    # stops after 19 rejections), so the tail's LENGTH is bounded, its STRUCTURE is asserted (`stalled_tail`), and every
    # iteration both runs have is compared, as is the minimum.
    n_host = int(st["MaxIt"])
    n_cmp = min(n_host, n_it)
    # WHERE the runs may part: only inside the stalled tail.  `stall` = first iteration from which the reference's objective
    # no longer moves (|fx - f_final| <= 1e-9 |f_final|); the first iteration the two runs do not share (n_cmp) must lie
    # beyond it, and whatever this run does past n_cmp are rejected steps that leave the objective where it was.
    ref_fx = np.asarray(ref["fx_trace"], dtype=float)[:n_it]
    moving = np.nonzero(np.abs(ref_fx - ref["f_final"]) > 1e-9 * abs(ref["f_final"]))[0]
    stall = int(moving[-1]) + 1 if moving.size else 0
    assert stall <= 12 and n_cmp >= stall + 10, (stall, n_cmp)     # 10 progress-making iterations, then >= 10 shared stalled ones

    def stalled_tail(n_own, beta, f_evals):
        """Past the shared iterations a run may only reject steps: beta x 4 per iteration, one evaluation each, and it must
        end within the half-ulp bound."""
        assert stall + 10 <= n_own <= stall + 30, (stall, n_own)
        b = np.asarray(beta, dtype=float).ravel()[:n_own]
        assert np.array_equal(b[stall + 1:], 4.0 * b[stall:-1]), b[stall:]
        if f_evals is not None:
            assert abs((int(f_evals) - ref["f_eval"]) - (n_own - n_it)) <= 1, (f_evals, n_own)

    stalled_tail(n_it, ref["beta_trace"][:n_it], ref["f_eval"])    # (the reference's own tail has that structure)
    stalled_tail(n_host, st["beta"], st["f_eval"])
    own_fx = np.asarray(st["fx"], dtype=float).ravel()[:n_host]
    assert np.all(np.abs(own_fx[n_cmp:] - ref["f_final"]) <= 1e-9 * abs(ref["f_final"]))
    assert np.allclose(st["fx"][:n_cmp], ref["fx_trace"][:n_cmp], rtol=1e-9, atol=0)
    assert np.allclose(st["beta"][:n_cmp], ref["beta_trace"][:n_cmp], rtol=1e-12, atol=0)
    assert abs(fx - ref["f_final"]) <= 1e-9 * abs(ref["f_final"])
    assert abs(np.linalg.norm(x) - ref["x_norm"]) <= 1e-11 * ref["x_norm"]
    dev = v.device_scg(dict(opts))
    x_d, f_d = dev(x0.copy())
    n_dev = int(dev.statistics["MaxIt"][0])
    stalled_tail(n_dev, np.asarray(dev.statistics["beta"])[:, 0], None)
    n_cmp = min(n_dev, n_it)
    assert n_cmp >= stall + 10
    assert np.all(np.abs(dev.statistics["fx"][n_cmp:n_dev, 0] - ref["f_final"]) <= 1e-9 * abs(ref["f_final"]))
    assert np.allclose(dev.statistics["fx"][:n_cmp, 0], ref["fx_trace"][:n_cmp], rtol=1e-9, atol=0)
    assert abs(f_d - ref["f_final"]) <= 1e-9 * abs(ref["f_final"])


@pytest.mark.parametrize("cfg", ["config1", "config2"])
def test_whole_optimisation_of_the_small_configs(cfg):
    """BASELINE configs[0] (OU, Euler) and configs[1] (Lorenz-63, RK4), t in [0, 10]: the reference's complete SCG run
    (52 / 57 iterations, 110 / 159 evaluations; tests/golden/scg_full_config{1,2}.json) against the host SCG here."""
    import json
    import os
    from conftest import GOLDEN_DIR
    ref = json.load(open(os.path.join(GOLDEN_DIR, f"scg_full_{cfg}.json")))
    n_it = ref["MaxIt_stat"]
    p = build_problem(ref["model"], ref["method"], ref["tf"], 0.01, None)
    v = p["vgp"]
    host = va.SCG(v.free_energy, v.gradient, {"max_it": ref["max_it"], "x_tol": 1e-6, "f_tol": 1e-8, "display": False})
    x, fx = host(v.initialization())
    st = host.statistics
    assert st["MaxIt"] == n_it and st["f_eval"] == ref["f_eval"]
    assert np.allclose(st["fx"][:n_it], ref["fx_trace"], rtol=1e-7, atol=0)
    assert abs(fx - ref["f_final"]) <= 1e-8 * abs(ref["f_final"])
    assert abs(np.linalg.norm(x) - ref["x_norm"]) <= 1e-7 * ref["x_norm"]
