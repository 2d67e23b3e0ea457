"""Shared helpers of the test-suite: build vgpa_amd objects from a golden fixture / from a seed."""
import io
import contextlib

import numpy as np

import vgpa_amd as va

SEED = 31415926535
MODEL_SETUP = {            # (ctor args, obs density, obs noise) used by tools/gen_golden.py
    "OU": (lambda d: (0.8, 1.0), 2, 0.04),
    "DW": (lambda d: (0.8, 1.0), 2, 0.04),
    "L63": (lambda d: ([10.0] * 3, [10.0, 28.0, 2.667]), 5, 2.0),
    "L96": (lambda d: ([4.0] * d, 8.0), 8, 1.0),
}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_model(name, dim_d=None, seed=SEED):
    args = MODEL_SETUP[name][0](dim_d or 40)
    cls = va.dynamical_systems[name]
    if name == "L96":
        return quiet(cls, *args, seed, dim_d or 40)
    return quiet(cls, *args, seed)


def build_problem(name, method, tf, dt=0.01, dim_d=None, seed=SEED, device=0, flags=0):
    """Same wiring as Simulation.setup/run (simulation.py:134-212) with vgpa_amd's classes."""
    model = make_model(name, dim_d, seed)
    model.make_trajectory(0.0, tf, dt)
    obs_t, obs_y, obs_noise = model.collect_obs(MODEL_SETUP[name][1], MODEL_SETUP[name][2], None)
    single = model.single_dim
    if single:
        m0 = model.sample_path[0] + 0.1 * model.rng.standard_normal()
        s0, mu0, tau0 = 0.2, 1.0, 0.5
    else:
        d = model.sample_path.shape[-1]
        m0 = model.sample_path[0] + 0.1 * model.rng.standard_normal(d)
        s0, mu0, tau0 = 0.2 * np.eye(d), np.ones(d), 0.5 * np.eye(d)
    fwd, bwd = va.FwdOde(dt, method, single), va.BwdOde(dt, method, single)
    lik = va.GaussianLikelihood(obs_y, obs_t, obs_noise, None, single)
    kl0 = va.PriorKL0(mu0, tau0, single)
    vgp = va.VarGP(model, m0, s0, fwd, bwd, lik, kl0, obs_y, obs_t, device=device, flags=flags)
    return dict(model=model, vgp=vgp, lik=lik, kl0=kl0, fwd=fwd, bwd=bwd, m0=m0, s0=s0, mu0=mu0, tau0=tau0,
                obs_t=obs_t, obs_y=obs_y, obs_noise=obs_noise)


def problem_from_golden(z, flags=0):
    """vgpa_amd objects wired on the fixture's own inputs (no random numbers drawn)."""
    name = str(z["model"])
    single = name in ("OU", "DW")
    d = 1 if single else int(np.asarray(z["m0"]).size)
    model = make_model(name, d)
    model.sample_path, model.time_window = z["sample_path"], z["time_window"]
    dt, method = float(z["dt"]), str(z["method"])
    m0 = float(z["m0"]) if single else z["m0"]
    s0 = float(z["s0"]) if single else z["s0"]
    mu0 = float(z["mu0"]) if single else z["mu0"]
    tau0 = float(z["tau0"]) if single else z["tau0"]
    obs_noise = float(z["obs_noise"]) if single else z["obs_noise"]
    obs_t = list(z["obs_t"])
    fwd, bwd = va.FwdOde(dt, method, single), va.BwdOde(dt, method, single)
    fwd.solver.flags = bwd.solver.flags = flags
    lik = va.GaussianLikelihood(z["obs_y"], obs_t, obs_noise, None, single)
    kl0 = va.PriorKL0(mu0, tau0, single)
    vgp = va.VarGP(model, m0, s0, fwd, bwd, lik, kl0, z["obs_y"], obs_t, flags=flags)
    return dict(model=model, vgp=vgp, lik=lik, kl0=kl0, fwd=fwd, bwd=bwd, single=single, d=d)


def split_x(x, n, d):
    if d == 1:
        return x[:n], x[n:]
    return x[:n * d * d].reshape(n, d, d), x[n * d * d:].reshape(n, d)
