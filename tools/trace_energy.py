"""Cycles per phase of the D <= 64 L96 energy kernel (k_energy_l96_r), summed over all waves of bench.py's workload.

Diagnostic build only:   VGPA_EXTRA_CFLAGS="-DVGPA_EXPERIMENTS -DVGPA_ENERGY_TRACE" python -m vgpa_amd.build && VGPA_ALLOW_DIAGNOSTIC=1 python tools/trace_energy.py
(the stamps are s_memtime reads at the phase boundaries; a build without the macro has none of them).
"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PHASES = ["entry + stage S_t", "Cholesky", "A fragments + G = A.L", "residuals (interior rows)", "boundary rows + sums",
          "L^-1", "dE/dm, dE/dS (X^T q X)", "<f>, stores"]


def main():
    from helpers import build_problem
    import torch  # noqa: F401  (one HIP runtime)
    import vgpa_amd as va
    from vgpa_amd import _lib
    d, n_pts, dt, B = 40, 1001, 0.01, int(os.environ.get("BATCH", "512"))
    p = build_problem("L96", "rk4", (n_pts - 1) * dt, dt, d)
    x0 = p["vgp"].initialization()
    e0 = float(p["kl0"](p["m0"], p["s0"]))
    ctx = va.Context("L96", "rk4", d, n_pts, dt, sigma=p["model"].sigma, theta=[8.0], m0=p["m0"], s0=p["s0"],
                     obs_t=p["obs_t"], obs_y=p["obs_y"], obs_noise=p["obs_noise"], e0=e0, batch=B)
    xb = np.stack([x0 + 0.05 * np.random.default_rng(1000 + i).standard_normal(x0.size) for i in range(B)])
    x_dev, g_dev = ctx.alloc(B * x0.size), ctx.alloc(B * x0.size)
    x_dev.upload(xb)
    lib = _lib.load()
    fn = lib.vgpa_debug_energy_trace
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 16)()
    for _ in range(3):
        ctx.sweep_enqueue(x_dev, g_dev)
        ctx.fetch_f()
    assert fn(buf, 1) == 0
    steps = 5
    for _ in range(steps):
        ctx.sweep_enqueue(x_dev, g_dev)
        ctx.fetch_f()
    assert fn(buf, 0) == 0
    waves = steps * B * n_pts
    per = [buf[i] / waves for i in range(8)]
    tot = sum(per)
    print(json.dumps({"cycles_per_wave_s_memtime": {PHASES[i]: round(per[i], 1) for i in range(8)}, "total": round(tot, 1),
                      "share": {PHASES[i]: round(per[i] / tot, 3) for i in range(8)}, "batch": B}, indent=1))


if __name__ == "__main__":
    main()
