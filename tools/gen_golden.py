#!/usr/bin/env python3
"""
Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the read-only reference at /root/reference (with an identity `numba.njit`
shim, because numba is not installed here and the decorated functions are pure
numpy) and dumps, for a set of short-grid cases, the INPUTS and OUTPUTS of the
hot path:

    FwdOde -> GaussianLikelihood -> model.energy -> BwdOde -> VarGP.gradient

to small `.npz` fixtures under tests/golden/.  It also writes scalar anchors of
the full-size BASELINE configurations to tests/golden/anchors.json.

Only data is written: no reference source, bytecode or pickled reference
objects ever leave this container.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--skip-full]
"""
import io
import os
import sys
import json
import tempfile
import contextlib

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 31415926535


def _install_shim():
    """Identity-njit shim: `@njit` and `@njit(fastmath=True)` both become no-ops."""
    shim = tempfile.mkdtemp(prefix="numba_shim_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as fh:
        fh.write("def njit(*a, **k):\n"
                 "    return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)\n")
    sys.path.insert(0, REF)
    sys.path.insert(0, shim)


_install_shim()
sys.dont_write_bytecode = True

from src.var_bayes.fwd_ode import FwdOde                      # noqa: E402
from src.var_bayes.bwd_ode import BwdOde                      # noqa: E402
from src.var_bayes.variational import VarGP                   # noqa: E402
from src.var_bayes.prior_kl0 import PriorKL0                  # noqa: E402
from src.var_bayes.gaussian_like import GaussianLikelihood    # noqa: E402
from src.dynamics.lorenz_96 import Lorenz96                   # noqa: E402
from src.dynamics.lorenz_63 import Lorenz63                   # noqa: E402
from src.dynamics.double_well import DoubleWell               # noqa: E402
from src.dynamics.ornstein_uhlenbeck import OrnsteinUhlenbeck  # noqa: E402


def build(model_name, method, tf, dt=0.01, dim_d=None, perturb=0.0, seed=SEED):
    """Replicates Simulation.setup/run wiring (simulation.py:134-212) up to VarGP."""
    with contextlib.redirect_stdout(io.StringIO()):
        if model_name == "OU":
            model, n_obs, r_obs = OrnsteinUhlenbeck(0.8, 1.0, seed), 2, 0.04
        elif model_name == "DW":
            model, n_obs, r_obs = DoubleWell(0.8, 1.0, seed), 2, 0.04
        elif model_name == "L63":
            model, n_obs, r_obs = Lorenz63([10.0] * 3, [10.0, 28.0, 2.667], seed), 5, 2.0
        elif model_name == "L96":
            d = dim_d or 40
            model, n_obs, r_obs = Lorenz96([4.0] * d, 8.0, seed, d), 8, 1.0
        else:
            raise ValueError(model_name)
    model.make_trajectory(0.0, tf, dt)
    obs_t, obs_y, obs_noise = model.collect_obs(n_obs, r_obs, None)
    single = model.single_dim
    if single:
        m0 = model.sample_path[0] + 0.1 * model.rng.standard_normal()
        s0 = 0.2
        mu0, tau0 = 1.0, 0.5
    else:
        dd = model.sample_path.shape[-1]
        m0 = model.sample_path[0] + 0.1 * model.rng.standard_normal(dd)
        s0 = 0.2 * np.eye(dd)
        mu0, tau0 = 1.0 * np.ones(dd), 0.5 * np.eye(dd)
    fwd = FwdOde(dt, method, single)
    bwd = BwdOde(dt, method, single)
    lik = GaussianLikelihood(obs_y, obs_t, obs_noise, None, single)
    kl0 = PriorKL0(mu0, tau0, single)
    vgp = VarGP(model, m0, s0, fwd, bwd, lik, kl0, obs_y, obs_t)
    x0 = vgp.initialization()
    x = x0.copy()
    if perturb > 0.0:
        x = x0 + perturb * np.random.default_rng(0).standard_normal(x0.size)
    return dict(model=model, vgp=vgp, lik=lik, kl0=kl0, fwd=fwd, bwd=bwd, x0=x0, x=x,
                m0=m0, s0=s0, mu0=mu0, tau0=tau0, obs_t=obs_t, obs_y=obs_y,
                obs_noise=obs_noise, dt=dt, tf=tf, method=method, name=model_name,
                n_obs=n_obs, r_obs=r_obs)


def evaluate(c):
    """Runs one sweep through the reference and returns every intermediate array."""
    model, vgp, lik, x = c["model"], c["vgp"], c["lik"], c["x"]
    n, d = vgp.dim_n, vgp.dim_d
    if d == 1:
        A, b = x[:vgp.dim_tot], x[vgp.dim_tot:]
    else:
        A, b = x[:vgp.dim_tot].reshape(n, d, d), x[vgp.dim_tot:].reshape(n, d)
    F = vgp.free_energy(x)
    g = vgp.gradient(x)
    out = vgp.arg_out
    mt, st = out["mt"], out["st"]
    Eobs = lik(mt, st)
    Esde, (Efx, Edf), (dEsde_dm, dEsde_ds, dEsde_dth, dEsde_dSig) = model.energy(A, b, mt, st, c["obs_t"])
    dEobs_dm, dEobs_ds, *_ = lik.gradients(mt, st)
    E0 = c["kl0"](c["m0"], c["s0"])
    return dict(A=A, b=b, F=F, grad=g, mt=mt, st=st, Eobs=Eobs, Esde=Esde, Efx=Efx, Edf=Edf,
                dEsde_dm=dEsde_dm, dEsde_ds=dEsde_ds, dEsde_dth=dEsde_dth, dEsde_dSig=dEsde_dSig,
                dEobs_dm=dEobs_dm, dEobs_ds=dEobs_ds, lamt=out["lamt"], psit=out["psit"], E0=E0)


def dump_case(tag, c, r):
    model = c["model"]
    np.savez_compressed(
        os.path.join(OUT, f"{tag}.npz"),
        # ---- inputs
        model=np.array(c["name"]), method=np.array(c["method"]), dt=np.array(c["dt"]),
        tf=np.array(c["tf"]), seed=np.array(SEED), n_obs=np.array(c["n_obs"]), r_obs=np.array(c["r_obs"]),
        theta=np.asarray(model.theta, dtype=float), sigma=np.asarray(model.sigma, dtype=float),
        inverse_sigma=np.asarray(model.inverse_sigma, dtype=float),
        time_window=model.time_window, sample_path=model.sample_path,
        m0=np.asarray(c["m0"]), s0=np.asarray(c["s0"]), mu0=np.asarray(c["mu0"]), tau0=np.asarray(c["tau0"]),
        obs_t=np.asarray(c["obs_t"], dtype=np.int64), obs_y=np.asarray(c["obs_y"]),
        obs_noise=np.asarray(c["obs_noise"], dtype=float),
        x0=c["x0"], x=c["x"],
        # ---- outputs
        F=np.array(r["F"]), grad=r["grad"], mt=r["mt"], st=r["st"], Eobs=np.array(r["Eobs"]),
        Esde=np.array(r["Esde"]), E0=np.array(r["E0"]), Efx=r["Efx"], Edf=r["Edf"],
        dEsde_dm=r["dEsde_dm"], dEsde_ds=r["dEsde_ds"],
        dEsde_dth=np.asarray(r["dEsde_dth"]), dEsde_dSig=np.asarray(r["dEsde_dSig"]),
        dEobs_dm=r["dEobs_dm"], dEobs_ds=r["dEobs_ds"], lamt=r["lamt"], psit=r["psit"])


# (tag, model, method, tf, dim_d, perturb)
SHORT_CASES = [
    ("ou_euler", "OU", "Euler", 1.0, None, 0.0),
    ("ou_heun_p", "OU", "Heun", 1.0, None, 0.05),
    ("ou_rk2_p", "OU", "RK2", 1.0, None, 0.05),
    ("ou_rk4", "OU", "RK4", 1.0, None, 0.0),
    ("ou_rk4_p", "OU", "RK4", 1.0, None, 0.05),
    ("dw_euler_p", "DW", "Euler", 1.0, None, 0.05),
    ("dw_rk4_p", "DW", "RK4", 1.0, None, 0.05),
    ("l63_euler_p", "L63", "Euler", 0.6, None, 0.05),
    ("l63_heun_p", "L63", "Heun", 0.6, None, 0.05),
    ("l63_rk2_p", "L63", "RK2", 0.6, None, 0.05),
    ("l63_rk4", "L63", "RK4", 0.6, None, 0.0),
    ("l63_rk4_p", "L63", "RK4", 0.6, None, 0.05),
    ("l96d12_euler_p", "L96", "Euler", 0.5, 12, 0.05),
    ("l96d12_heun_p", "L96", "Heun", 0.5, 12, 0.05),
    ("l96d12_rk2_p", "L96", "RK2", 0.5, 12, 0.05),
    ("l96d12_rk4_p", "L96", "RK4", 0.5, 12, 0.05),
    ("l96d17_rk4_p", "L96", "RK4", 0.3, 17, 0.05),
    ("l96d40_rk2_p", "L96", "RK2", 0.25, 40, 0.05),
    ("l96d40_rk4", "L96", "RK4", 0.25, 40, 0.0),
    ("l96d40_rk4_p", "L96", "RK4", 0.25, 40, 0.05),
]

# Full-size anchors: (tag, model, method, tf, dim_d)
FULL_CASES = [
    ("ou_euler_full", "OU", "Euler", 10.0, None),
    ("ou_rk4_full", "OU", "RK4", 10.0, None),
    ("dw_rk4_full", "DW", "RK4", 10.0, None),
    ("l63_rk4_full", "L63", "RK4", 10.0, None),
    ("l96d40_rk2_tf4", "L96", "RK2", 4.0, 40),
    ("l96d40_rk4_tf4", "L96", "RK4", 4.0, 40),
    ("l96d40_rk4_full", "L96", "RK4", 10.0, 40),
]


def anchors_of(c, r):
    g = r["grad"]
    return dict(model=c["name"], method=c["method"], tf=c["tf"], dt=c["dt"],
                Np=int(c["vgp"].dim_n), D=int(c["vgp"].dim_d), len_x=int(c["x"].size),
                F=float(r["F"]), grad_norm=float(np.linalg.norm(g)), grad_sum=float(np.sum(g)),
                grad_absmax=float(np.abs(g).max()),
                Esde=float(r["Esde"]), Eobs=float(r["Eobs"]), E0=float(r["E0"]),
                mt_last=np.atleast_1d(r["mt"][-1]).tolist()[:8],
                st_trace_last=float(np.trace(np.atleast_2d(r["st"][-1]))),
                st_fro=float(np.linalg.norm(np.asarray(r["st"]).ravel())),
                lam0=np.atleast_1d(r["lamt"][0]).tolist()[:8],
                psi_fro=float(np.linalg.norm(np.asarray(r["psit"]).ravel())),
                x0_sum=float(np.sum(c["x0"])), x0_norm=float(np.linalg.norm(c["x0"])),
                n_obs_pts=int(len(c["obs_t"])),
                obs_y_sum=float(np.sum(c["obs_y"])), path_sum=float(np.sum(c["model"].sample_path)))


def dump_host_terms():
    """Inputs / outputs of the two host-side gradient helpers nothing on the hot path consumes:
    PriorKL0.gradients (prior_kl0.py:94-175) and the 1-D dEobs_dr (gaussian_like.py:154-196)."""
    rng = np.random.default_rng(2718)
    out = {}
    # 1-D KL0
    mu0, tau0, m0, s0, lam0, psi0 = 1.0, 0.5, 0.7, 0.2, -0.3, 0.11
    g = PriorKL0(mu0, tau0, True).gradients(m0, s0, lam0, psi0)
    out.update(kl1_in=np.array([mu0, tau0, m0, s0, lam0, psi0]), kl1_dm0=g[0], kl1_ds0=g[1])
    # n-D KL0
    d = 5
    q = rng.standard_normal((d, d))
    tau = 0.5 * np.eye(d) + 0.05 * (q + q.T) / 2.0
    q = rng.standard_normal((d, d))
    s0n = 0.2 * np.eye(d) + 0.02 * (q + q.T) / 2.0
    mu, m0n, lam = rng.standard_normal(d), rng.standard_normal(d), rng.standard_normal(d)
    q = rng.standard_normal((d, d))
    psi = (q + q.T) / 2.0
    g = PriorKL0(mu, tau, False).gradients(m0n, s0n, lam, psi)
    out.update(kln_mu0=mu, kln_tau0=tau, kln_m0=m0n, kln_s0=s0n, kln_lam0=lam, kln_psi0=psi, kln_dm0=g[0], kln_ds0=g[1])
    # 1-D dEobs_dr
    n = 40
    m, s = rng.standard_normal(n), 0.1 + rng.random(n)
    obs_t = [3, 9, 17, 31]
    obs_y = rng.standard_normal(len(obs_t))
    lik = GaussianLikelihood(obs_y, obs_t, 0.04, None, True)
    dm, ds, dr = lik.gradients(m, s)
    out.update(obs_m=m, obs_s=s, obs_t=np.asarray(obs_t), obs_y=obs_y, obs_noise=0.04, obs_dm=dm, obs_ds=ds, obs_dr=dr)
    np.savez_compressed(os.path.join(OUT, "host_terms.npz"), **out)
    print("host_terms        written")


def main():
    os.makedirs(OUT, exist_ok=True)
    skip_full = "--skip-full" in sys.argv
    if "--host-terms-only" in sys.argv:
        dump_host_terms()
        return
    dump_host_terms()
    anchors = {}
    for tag, name, method, tf, dd, pert in SHORT_CASES:
        c = build(name, method, tf, dim_d=dd, perturb=pert)
        r = evaluate(c)
        dump_case(tag, c, r)
        anchors[tag] = anchors_of(c, r)
        print(f"{tag:18s} Np={c['vgp'].dim_n:5d} D={c['vgp'].dim_d:3d}  F={r['F']:.12e}  |g|={np.linalg.norm(r['grad']):.12e}")
    if not skip_full:
        for tag, name, method, tf, dd in FULL_CASES:
            for pert, sfx in ((0.0, ""), (0.05, "_p")):
                c = build(name, method, tf, dim_d=dd, perturb=pert)
                r = evaluate(c)
                anchors[tag + sfx] = anchors_of(c, r)
                print(f"{tag + sfx:18s} Np={c['vgp'].dim_n:5d} D={c['vgp'].dim_d:3d}  F={r['F']:.12e}  |g|={np.linalg.norm(r['grad']):.12e}")
        with open(os.path.join(OUT, "anchors.json"), "w") as fh:
            json.dump(anchors, fh, indent=1, sort_keys=True)
    else:
        with open(os.path.join(OUT, "anchors_short.json"), "w") as fh:
            json.dump(anchors, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
