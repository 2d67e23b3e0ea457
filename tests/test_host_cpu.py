"""CPU suite: host logic of vgpa_amd (no GPU compute) and the C-ABI library surface."""
import os
import re
import json

import numpy as np
import pytest

import vgpa_amd as va
from conftest import ROOT, GOLDEN_DIR, rel_err
from helpers import build_problem, problem_from_golden, quiet


def test_library_builds_loads_and_exports_every_declared_symbol():
    lib = va.load()
    header = open(os.path.join(ROOT, "include", "vgpa_hip.h")).read()
    declared = set(re.findall(r"\b(vgpa_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vgpa_hip.h but not exported"
    from vgpa_amd._lib import SYMBOLS
    assert set(SYMBOLS) == declared
    assert lib.vgpa_abi_version() == 2          # version 2, and NOT a diagnostic build (bit 16, vgpa_hip.h VGPA_ABI_DIAGNOSTIC_BUILD)


def test_python_constants_agree_with_the_header():
    """Flags, fetch selectors and status codes are plain integers on both sides of the ctypes boundary."""
    from vgpa_amd import _lib
    header = open(os.path.join(ROOT, "include", "vgpa_hip.h")).read()
    enums = {k: int(v) for k, v in re.findall(r"\b(VGPA_[A-Z0-9_]+)\s*=\s*(-?\d+)", header)}
    for name in ("FORCE_GENERIC", "STREAM_LARGE_D", "LIBRARY_GEMM", "SYM_UNITS", "KEEP_PSI", "MATERIALIZE"):
        assert getattr(_lib, "FLAG_" + name) == enums["VGPA_FLAG_" + name], name
    fetch = {"mt": "MT", "st": "ST", "lamt": "LAMT", "psit": "PSIT", "Efx": "EFX", "Edf": "EDF", "dEsde_dm": "DESDE_DM",
             "dEsde_ds": "DESDE_DS", "Esde_t": "ESDE_T"}
    for key, sel in fetch.items():
        assert _lib.FETCH_IDS[key] == enums["VGPA_FETCH_" + sel], key
    assert len({enums[k] for k in enums if k.startswith("VGPA_FLAG_")}) == 6          # distinct bits
    assert _lib.SHARD_OPT_GATHER_CHUNKS == enums["VGPA_SHARD_OPT_GATHER_CHUNKS"]
    assert _lib.SHARD_OPT_TIMEOUT_MS == enums["VGPA_SHARD_OPT_TIMEOUT_MS"]


def test_no_silent_cpu_fallback():
    """Without a HIP device every compute entry point must fail loudly."""
    if va.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        va.FwdOde(0.01, "rk4", False)(np.zeros((5, 3, 3)), np.zeros((5, 3)), np.zeros(3), np.eye(3), np.eye(3))
    model = quiet(va.OrnsteinUhlenbeck, 0.8, 1.0, 1)
    model.make_trajectory(0.0, 0.1, 0.01)
    with pytest.raises(RuntimeError, match="no HIP device"):
        model.energy(np.ones(11), np.ones(11), np.ones(11), np.ones(11), [])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vgpa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "vgpa_oracle" not in src, f
                assert "/root/reference" not in src, f


def test_stepper_registry_and_errors():
    assert set(va.num_integration) == {"euler", "heun", "rk2", "rk4"}
    assert set(va.dynamical_systems) == {"DW", "OU", "L63", "L96"}
    with pytest.raises(ValueError):
        va.FwdOde(0.0, "rk4", True)
    with pytest.raises(ValueError):
        va.BwdOde(-0.1, "euler", True)
    with pytest.raises(ValueError):
        va.FwdOde(0.01, "leapfrog", True)
    with pytest.raises(ValueError):
        va.Euler(0.0, True)
    assert isinstance(va.FwdOde(0.01, "RK4", False).solver, va.RungeKutta4)   # case-insensitive like the reference
    with pytest.raises(ValueError):
        quiet(va.OrnsteinUhlenbeck, -1.0, 1.0)
    with pytest.raises(ValueError):
        quiet(va.Lorenz96, [4.0] * 5, 8.0, 1, 5)
    with pytest.raises(RuntimeError):
        quiet(va.Lorenz63, [-1.0, 1.0, 1.0], [10.0, 28.0, 2.667])


def test_input_generators_reproduce_reference_streams(golden):
    """make_trajectory / collect_obs / m0 draw / initialization consume the seeded stream like the reference."""
    name = str(golden["model"])
    d = None if name in ("OU", "DW", "L63") else int(golden["m0"].size)
    p = build_problem(name, str(golden["method"]), float(golden["tf"]), float(golden["dt"]), d)
    assert np.array_equal(p["model"].time_window, golden["time_window"])
    assert rel_err(p["model"].sample_path, golden["sample_path"]) < 1e-14
    assert np.array_equal(np.asarray(p["obs_t"]), golden["obs_t"])
    assert rel_err(p["obs_y"], golden["obs_y"]) < 1e-14
    assert rel_err(p["m0"], golden["m0"]) < 1e-14
    x0 = p["vgp"].initialization()
    assert x0.shape == golden["x0"].shape
    assert rel_err(x0, golden["x0"]) < 1e-13
    assert rel_err(p["kl0"](p["m0"], p["s0"]), golden["E0"]) < 1e-13
    assert rel_err(p["model"].inverse_sigma, golden["inverse_sigma"]) < 1e-15


def test_full_size_inputs_match_anchors():
    """The BASELINE-size inputs (Np=1001) regenerate identically (checked through scalar anchors)."""
    anchors = json.load(open(os.path.join(GOLDEN_DIR, "anchors.json")))
    for tag, name, method, d in (("ou_euler_full", "OU", "Euler", None), ("l63_rk4_full", "L63", "RK4", None),
                                 ("l96d40_rk4_full", "L96", "RK4", 40)):
        a = anchors[tag]
        p = build_problem(name, method, a["tf"], a["dt"], d)
        x0 = p["vgp"].initialization()
        assert x0.size == a["len_x"]
        assert abs(np.sum(x0) - a["x0_sum"]) <= 1e-11 * abs(a["x0_sum"])
        assert abs(np.linalg.norm(x0) - a["x0_norm"]) <= 1e-12 * a["x0_norm"]
        assert abs(np.sum(p["obs_y"]) - a["obs_y_sum"]) <= 1e-11 * abs(a["obs_y_sum"])
        assert abs(np.sum(p["model"].sample_path) - a["path_sum"]) <= 1e-11 * abs(a["path_sum"])
        assert abs(float(p["kl0"](p["m0"], p["s0"])) - a["E0"]) <= 1e-12 * abs(a["E0"])


def test_scg_minimises_sphere_and_rosenbrock():
    """Same two problems as the reference's src/tests/test_scg.py (which cannot pass there, see its header)."""
    def sphere(x):
        return float(np.sum(x ** 2))

    def dsphere(x, eval_fun=False):
        return 2.0 * x

    opt = va.SCG(sphere, dsphere, {"max_it": 500, "x_tol": 1e-8, "f_tol": 1e-10})
    x, fx = opt(np.random.default_rng(1).standard_normal(25) * 3)
    assert fx < 1e-10 and np.allclose(x, 0.0, atol=1e-5)

    def rosen(x):
        return float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2))

    def drosen(x, eval_fun=False):
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2.0 * (1.0 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return g

    opt = va.SCG(rosen, drosen, {"max_it": 5000, "x_tol": 1e-10, "f_tol": 1e-12})
    x, fx = opt(np.array([-1.2, 1.0, -0.5, 0.8]))
    assert fx < 1e-8 and np.allclose(x, 1.0, atol=1e-3)
    assert opt.statistics["f_eval"] > 0 and opt.statistics["df_eval"] > 0


def test_vargp_wiring_without_gpu(golden):
    p = problem_from_golden(golden)
    v = p["vgp"]
    assert v.dim_n == golden["time_window"].size
    assert v.dim_tot == v.dim_n * v.dim_d ** 2
    assert set(v.arg_out) >= {"m0", "s0"}


def test_host_side_gradient_helpers_match_the_reference():
    """PriorKL0.gradients and the 1-D dEobs_dr against vectors generated from the reference (tools/gen_golden.py);
    both are host numpy in the reference too and sit outside the hot path."""
    from conftest import load_golden
    z = load_golden("host_terms")
    mu0, tau0, m0, s0, lam0, psi0 = z["kl1_in"]
    g = va.PriorKL0(mu0, tau0, True).gradients(m0, s0, lam0, psi0)
    assert np.allclose([g[0], g[1]], [z["kl1_dm0"], z["kl1_ds0"]], rtol=1e-13, atol=0)
    g = va.PriorKL0(z["kln_mu0"], z["kln_tau0"], False).gradients(z["kln_m0"], z["kln_s0"], z["kln_lam0"], z["kln_psi0"])
    assert np.allclose(g[0], z["kln_dm0"], rtol=1e-12, atol=1e-14) and np.allclose(g[1], z["kln_ds0"], rtol=1e-12, atol=1e-14)


def test_host_scg_reproduces_the_references_recorded_optimisation_on_the_oracle_objective():
    """BASELINE configs[0] (OU, Euler, t in [0, 10]): the reference's complete SCG run (52 iterations, 110 objective
    evaluations, tests/golden/scg_full_config1.json, written by tools/gen_scg_anchor.py) replayed on the CPU: the lock-step
    engine with host vectors drives the numpy oracle's objective and must follow the recorded trace."""
    import json
    import os
    from conftest import GOLDEN_DIR
    from helpers import build_problem
    from oracle import vgpa_oracle as vo
    ref = json.load(open(os.path.join(GOLDEN_DIR, "scg_full_config1.json")))
    p = build_problem(ref["model"], ref["method"], ref["tf"], 0.01, None)
    z = dict(model="OU", method=ref["method"], dt=0.01, theta=1.0, sigma=0.8, m0=p["m0"], s0=p["s0"], mu0=p["mu0"],
             tau0=p["tau0"], obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"],
             time_window=p["model"].time_window)
    prob = vo.Problem.from_fixture({k: np.asarray(val) for k, val in z.items()})
    cache = {}

    def f_cpu(x):
        f, cache["state"] = vo.free_energy(prob, x)
        return f

    def df_cpu(x, eval_fun=False):
        if eval_fun:
            f_cpu(x)
        return vo.gradient(prob, x, cache["state"])

    opt = va.SCG(f_cpu, df_cpu, {"max_it": ref["max_it"], "x_tol": 1e-6, "f_tol": 1e-8, "display": False})
    x, fx = opt(p["vgp"].initialization())
    st = opt.statistics
    n_it = ref["MaxIt_stat"]
    assert st["MaxIt"] == n_it and st["f_eval"] == ref["f_eval"]
    assert np.allclose(st["fx"][:n_it], ref["fx_trace"], rtol=1e-8, atol=0)
    assert np.allclose(st["beta"][:n_it], ref["beta_trace"], rtol=1e-12, atol=0)
    assert abs(fx - ref["f_final"]) <= 1e-9 * abs(ref["f_final"])
    assert abs(np.linalg.norm(x) - ref["x_norm"]) <= 1e-8 * ref["x_norm"]


def test_lock_step_engine_freezes_finished_problems():
    """Two host problems of different difficulty in one lock-step run == the two runs alone."""
    from vgpa_amd import scg as S

    class Two(S._HostVectors):
        def __init__(self):
            super().__init__(None, None)
            self.B = 2
            self.scale = np.array([1.0, 50.0])

        def _f(self, x):
            return np.array([np.sum(self.scale[k] * x[k] ** 2) + np.sum(x[k] ** 4) for k in range(2)])

        def _g(self, x):
            return np.stack([2.0 * self.scale[k] * x[k] + 4.0 * x[k] ** 3 for k in range(2)])

        def value_and_gradient(self, x, g, st):
            g[...] = self._g(x)
            return self._f(x)

        def probe_gradient(self, x, g, st):
            g[...] = self._g(x)

        def value(self, x, st):
            return self._f(x)

    x0 = np.random.default_rng(3).standard_normal((2, 6))
    st = S._new_stats(200, 2)
    x, f = S._lock_step(Two(), x0, 200, 1e-9, 1e-12, False, st)
    assert np.all(f < 1e-8) and np.abs(x).max() < 1e-3
    assert st["MaxIt"][0] != st["MaxIt"][1]               # they stop at different iterations ...
    for k in range(2):                                    # ... and each equals its own single run
        def fk(v, k=k):
            return float(np.sum([1.0, 50.0][k] * v ** 2) + np.sum(v ** 4))

        def gk(v, eval_fun=False, k=k):
            return 2.0 * [1.0, 50.0][k] * v + 4.0 * v ** 3
        solo = va.SCG(fk, gk, {"max_it": 200, "x_tol": 1e-9, "f_tol": 1e-12})
        xs, fs = solo(x0[k])
        assert solo.statistics["MaxIt"] == st["MaxIt"][k]
        assert np.allclose(xs, x[k], rtol=0, atol=1e-12)


def test_device_contexts_are_rebuilt_when_a_baked_in_input_changes(monkeypatch):
    """theta / sigma / the observation noise / (m0, s0) are settable on the reference's objects; a context created from
    older values must not be reused.  No GPU: the Context class is replaced by a recorder."""
    import vgpa_amd.variational as V
    import vgpa_amd.likelihood as L
    from helpers import build_problem
    made = []

    class Recorder:
        def __init__(self, *a, **k):
            self.args, self.kw, self.closed = a, k, False
            made.append(self)

        def close(self):
            self.closed = True

        def set_prior_energy(self, e0):
            self.e0 = e0

    monkeypatch.setattr(V, "Context", Recorder)
    monkeypatch.setattr(L, "Context", Recorder)
    p = build_problem("L96", "RK4", 0.2, 0.01, 12)
    v = p["vgp"]
    c0 = v._context()
    assert v._context() is c0 and len(made) == 1
    assert c0.kw["obs_h"] is None                          # default operator -> the library's diagonal fast path
    # the prior is NOT baked in: E0 follows kl0.mu0 / kl0.tau0 on the next call, same context
    e0_first = c0.e0
    assert e0_first == float(p["kl0"](p["m0"], p["s0"]))
    p["kl0"].mu0 = p["kl0"].mu0 + 0.25
    assert v._context() is c0 and c0.e0 == float(p["kl0"](p["m0"], p["s0"])) and c0.e0 != e0_first
    # large arrays are identified by a digest of the whole buffer, not by a few probes: a permutation is a change
    big = np.arange(70000, dtype=float)
    k_a = V.VarGP._fingerprint({"a": big})
    big2 = big.copy(); big2[[5, 60000]] = big2[[60000, 5]]
    assert V.VarGP._fingerprint({"a": big2}) != k_a and V.VarGP._fingerprint({"a": big.copy()}) == k_a
    p["model"].theta = 9.0
    c1 = v._context()
    assert c1 is not c0 and c0.closed and c1.kw["theta"][0] == 9.0
    p["lik"].noise = 2.0 * np.asarray(p["lik"].noise)
    c2 = v._context()
    assert c2 is not c1 and c1.closed and np.array_equal(c2.kw["obs_noise"], np.asarray(p["lik"].noise))
    v.output["m0"] = v.output["m0"] + 1.0
    assert v._context() is not c2
    v.invalidate()
    assert v._ctx is None
    # the likelihood's own operator-level context follows its noise setter
    lik = p["lik"]
    k0 = lik._context(21, 12)
    assert lik._context(21, 12) is k0
    lik.noise = 3.0 * np.asarray(lik.noise)
    assert k0.closed and lik._context(21, 12) is not k0
    # an explicit operator is passed through
    h = np.eye(12); h[0, 1] = 0.5
    lik2 = va.GaussianLikelihood(p["obs_y"], p["obs_t"], p["obs_noise"], h, False)
    assert np.array_equal(lik2._context(21, 12).kw["obs_h"], h)
