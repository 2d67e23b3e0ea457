"""
CPU suite: pins the numpy oracle to the golden vectors captured from the real reference
(tools/gen_golden.py).  Every stage of the sweep is compared, stage by stage AND end to end.
"""
import numpy as np
import pytest

from conftest import rel_err
from oracle import vgpa_oracle as vo

TOL = 1e-11


def _problem(z):
    return vo.Problem.from_fixture(z)


def test_forward_sweep(golden):
    p = _problem(golden)
    a, b = p.split(golden["x"])
    mt, st = vo.solve_fwd(p.method, p.dt, p.single_dim, a, b, p.m0, p.s0, p.sigma)
    assert rel_err(mt, golden["mt"]) < TOL
    assert rel_err(st, golden["st"]) < TOL


def test_energy_terms(golden):
    p = _problem(golden)
    a, b = p.split(golden["x"])
    for faithful in ((True, False) if p.model == "L96" else (True,)):
        esde, (ef, edf), (dm, ds, dth, dsig) = vo.model_energy(p, a, b, golden["mt"], golden["st"], faithful)
        tol = TOL if faithful else 1e-9
        assert rel_err(esde, golden["Esde"]) < tol
        assert rel_err(ef, golden["Efx"]) < tol
        assert rel_err(edf, golden["Edf"]) < tol
        assert rel_err(dm, golden["dEsde_dm"]) < tol
        assert rel_err(ds, golden["dEsde_ds"]) < tol
        assert rel_err(dth, golden["dEsde_dth"]) < tol
        assert rel_err(dsig, golden["dEsde_dSig"]) < tol


def test_observation_terms(golden):
    p = _problem(golden)
    assert rel_err(vo.eobs(p, golden["mt"], golden["st"]), golden["Eobs"]) < TOL
    jm, js = vo.eobs_gradients(p, golden["mt"], golden["st"])
    assert rel_err(jm, golden["dEobs_dm"]) < TOL
    assert rel_err(js, golden["dEobs_ds"]) < TOL
    assert rel_err(vo.kl0(p), golden["E0"]) < TOL


def test_backward_sweep(golden):
    p = _problem(golden)
    a, _ = p.split(golden["x"])
    lam, psi = vo.solve_bwd(p.method, p.dt, p.single_dim, a, golden["dEsde_dm"], golden["dEsde_ds"],
                            golden["dEobs_dm"], golden["dEobs_ds"])
    assert rel_err(lam, golden["lamt"]) < TOL
    assert rel_err(psi, golden["psit"]) < TOL


def test_full_sweep(golden):
    p = _problem(golden)
    f, g, state = vo.sweep(p, golden["x"], faithful=True)
    assert abs(f - float(golden["F"])) <= TOL * abs(float(golden["F"]))
    assert rel_err(g, golden["grad"]) < TOL
    assert rel_err(state["lamt"], golden["lamt"]) < TOL
    assert rel_err(state["psit"], golden["psit"]) < TOL


def test_full_sweep_lean_matches(golden):
    p = _problem(golden)
    if p.model != "L96":
        pytest.skip("lean mode only differs for L96")
    f, g, _ = vo.sweep(p, golden["x"], faithful=False)
    assert abs(f - float(golden["F"])) <= 1e-10 * abs(float(golden["F"]))
    assert rel_err(g, golden["grad"]) < 1e-9


def test_trapz_segments_equal_global():
    rng = np.random.default_rng(5)
    fx = rng.standard_normal((501, 3))
    obs = [10, 50, 77, 400]
    a = vo.my_trapz(fx, 0.01, obs)
    b = vo.my_trapz(fx, 0.01, None)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-14)


def test_l96_flat_roll_quirk():
    """Q1: the sigma-point matrix is rolled as a flat array (rows leak into each other)."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 12))
    flat = vo.l96_drift(x, 8.0)
    rowwise = np.stack([vo.l96_drift(r, 8.0) for r in x])
    assert not np.allclose(flat, rowwise)
    assert np.allclose(flat[:, 2:-1], rowwise[:, 2:-1])
