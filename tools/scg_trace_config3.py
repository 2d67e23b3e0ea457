import sys, os, io, json, contextlib
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import vgpa_amd as va
from helpers import build_problem
p = build_problem("L96", "RK4", 10.0, 0.01, 40)
v = p["vgp"]; x0 = v.initialization()
res = {}
for name, mk in (("host_scg", lambda: va.SCG(v.free_energy, v.gradient, {"max_it": 3, "x_tol": 1e-6, "f_tol": 1e-8, "display": False})),
                 ("device_scg", lambda: v.device_scg({"max_it": 3, "x_tol": 1e-6, "f_tol": 1e-8, "display": False}))):
    opt = mk()
    with contextlib.redirect_stdout(io.StringIO()):
        x, fx = opt(x0.copy())
    st = opt.statistics
    res[name] = {"fx_trace": [float(a) for a in np.asarray(st["fx"])[:3].ravel()], "beta_trace": [float(a) for a in np.asarray(st["beta"])[:3].ravel()],
                 "f_final": float(fx), "x_norm": float(np.linalg.norm(x))}
print(json.dumps(res))
