"""Runs the REFERENCE (imported from /root/reference with the identity-njit shim, build container only) through three SCG
iterations at BASELINE configs[2] (Lorenz-96, D=40, RK4, Np=1001) and prints the trace that tests/golden/scg_trace_config3.json
holds: `PYTHONDONTWRITEBYTECODE=1 python tools/gen_scg_anchor.py > tests/golden/scg_trace_config3.json` (about two minutes).
With the argument `full` it runs the optimiser to its own termination (max_it = 500; 30 iterations, 50 objective
evaluations, about nine minutes) -> tests/golden/scg_full_config3.json."""
import sys, os, io, json, time, contextlib, tempfile
import numpy as np
shim = tempfile.mkdtemp(prefix="numba_shim_"); os.makedirs(os.path.join(shim, "numba"))
open(os.path.join(shim, "numba", "__init__.py"), "w").write("def njit(*a, **k):\n    return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)\n")
sys.path.insert(0, "/root/reference"); sys.path.insert(0, shim); sys.dont_write_bytecode = True
from src.var_bayes.fwd_ode import FwdOde
from src.var_bayes.bwd_ode import BwdOde
from src.var_bayes.variational import VarGP
from src.var_bayes.prior_kl0 import PriorKL0
from src.var_bayes.gaussian_like import GaussianLikelihood
from src.dynamics.lorenz_96 import Lorenz96
from src.numerics.optim_scg import SCG
D = 40
with contextlib.redirect_stdout(io.StringIO()):
    model = Lorenz96([4.0] * D, 8.0, 31415926535)
    model.make_trajectory(0.0, 10.0, 0.01)
    obs_t, obs_y, R = model.collect_obs(8, 1.0, None)
m0 = model.sample_path[0] + 0.1 * model.rng.standard_normal(D); S0 = 0.2 * np.eye(D)
v = VarGP(model, m0, S0, FwdOde(0.01, "RK4", False), BwdOde(0.01, "RK4", False),
          GaussianLikelihood(obs_y, obs_t, R, None, False), PriorKL0(np.ones(D), 0.5 * np.eye(D), False), obs_y, obs_t)
x0 = v.initialization()
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"
N_IT, N_KEEP = (500, 40) if FULL else (3, 3)
opt = SCG(v.free_energy, v.gradient, {"max_it": N_IT, "x_tol": 1e-6, "f_tol": 1e-8, "display": False})
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    x, fx = opt(x0.copy())
st = opt.stats
print(json.dumps({"MaxIt_stat": int(st["MaxIt"]), "f_eval": float(st["f_eval"]), "reference_scg_max_it": N_IT,
                  "seconds": time.perf_counter() - t0, "fx_trace": [float(a) for a in st["fx"][:N_KEEP]],
                  "beta_trace": [float(a) for a in st["beta"][:N_KEEP]], "f_final": float(fx), "x_norm": float(np.linalg.norm(x))}))
