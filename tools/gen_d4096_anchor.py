#!/usr/bin/env python3
"""Oracle anchors of the fused sweep at the matrix size of BASELINE configs[4] (Lorenz-96, D = 4096, RK4) on a 4-point grid.

    python tools/gen_d4096_anchor.py            # ~10 minutes on 8 cores, ~6 GB; writes tests/golden/anchors_d4096.json

The reference itself cannot run at this size (its unscented transform builds a (D + D^2)^2 covariance), so the
numbers come from oracle/vgpa_oracle.py in lean mode -- the restatement that tests/test_oracle_golden.py pins to the reference
at D <= 40.  The inputs are the ones tests/test_large_d.py::test_native_sharded_fused_sweep_at_config5_matrix_size builds
(`inputs()` below is imported by that test): the anchors replace that test's former comparison of 2 / 8 ranks with the 1-rank run
of the same kernels.  Only scalars and a few sampled entries are stored (the gradient itself is 0.5 GB)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

D, N_PTS, METHOD, DT = 4096, 4, "rk4", 0.01
SAMPLES = [(0, 0), (0, 1), (1, 0), (17, 17), (17, 4000), (511, 512), (512, 511), (2047, 2048), (4095, 0), (4095, 4095),
           (1234, 3210), (3210, 1234)]


def inputs(d=D, n=N_PTS):
    """(x, m0, s0, sigma_diag, obs_t, obs_y, r_diag) -- seeded, the same on every machine."""
    rng = np.random.default_rng(5)
    m0 = 8.0 + rng.standard_normal(d)
    a = 8.0 * np.eye(d)[None] + (0.05 / np.sqrt(d)) * rng.standard_normal((n, d, d))
    b = 8.0 * m0[None] + rng.standard_normal((n, d))
    x = np.concatenate((a.ravel(), b.ravel()))
    obs_t = np.array([2], dtype=np.int64)
    obs_y = 8.0 + rng.standard_normal((1, d))
    return x, m0, 0.2 * np.eye(d), np.full(d, 4.0), obs_t, obs_y, np.ones(d)


def inputs_grid(d, n):
    """The same generator on a longer grid (BASELINE configs[3]'s matrix size with a real chunk of the time-chunked sweep): an
    observation every fourth grid point."""
    rng = np.random.default_rng(5)
    m0 = 8.0 + rng.standard_normal(d)
    a = 8.0 * np.eye(d)[None] + (0.05 / np.sqrt(d)) * rng.standard_normal((n, d, d))
    b = 8.0 * m0[None] + rng.standard_normal((n, d))
    x = np.concatenate((a.ravel(), b.ravel()))
    obs_t = np.arange(2, n - 1, 4, dtype=np.int64)
    obs_y = 8.0 + rng.standard_normal((obs_t.size, d))
    return x, m0, 0.2 * np.eye(d), np.full(d, 4.0), obs_t, obs_y, np.ones(d)


def main():
    global D, N_PTS, SAMPLES
    from oracle import vgpa_oracle as vo
    if len(sys.argv) > 2:              # python tools/gen_d4096_anchor.py 1024 33  ->  tests/golden/anchors_d1024_np33.json
        D, N_PTS = int(sys.argv[1]), int(sys.argv[2])
        SAMPLES = [(i % D, j % D) for (i, j) in SAMPLES]
        x, m0, s0, sig, obs_t, obs_y, rdiag = inputs_grid(D, N_PTS)
    else:
        x, m0, s0, sig, obs_t, obs_y, rdiag = inputs()
    p = vo.Problem(model="L96", method=METHOD, dt=DT, theta=8.0, sigma=np.diag(sig), m0=m0, s0=s0, mu0=np.ones(D),
                   tau0=0.5 * np.eye(D), obs_t=obs_t, obs_y=obs_y, obs_noise=np.diag(rdiag), n_pts=N_PTS, dim_d=D)
    t0 = time.perf_counter()
    f, g, st = vo.sweep(p, x, faithful=False)
    secs = time.perf_counter() - t0
    ga, gb = g[:N_PTS * D * D].reshape(N_PTS, D, D), g[N_PTS * D * D:].reshape(N_PTS, D)
    out = {
        "what": f"oracle/vgpa_oracle.py (lean mode) sweep of Lorenz-96 D={D} RK4 on a {N_PTS}-point grid; inputs = tools/gen_d4096_anchor.py::"
                + ("inputs()" if len(sys.argv) <= 2 else f"inputs_grid({D}, {N_PTS})"),
        "D": D, "Np": N_PTS, "method": METHOD, "dt": DT, "oracle_seconds": secs,
        "F_minus_E0": float(st["Esde"] + st["Eobs"]), "Esde": float(st["Esde"]), "Eobs": float(st["Eobs"]),
        "grad_a_fro": [float(np.linalg.norm(ga[t])) for t in range(N_PTS)],
        "grad_a_absmax": [float(np.abs(ga[t]).max()) for t in range(N_PTS)],
        "grad_b_norm": [float(np.linalg.norm(gb[t])) for t in range(N_PTS)],
        "grad_b_absmax": [float(np.abs(gb[t]).max()) for t in range(N_PTS)],
        "samples_ij": SAMPLES,
        "grad_a_samples": [[float(ga[t][i, j]) for (i, j) in SAMPLES] for t in range(N_PTS)],
        "grad_b_first8": [[float(v) for v in gb[t][:8]] for t in range(N_PTS)],
        "st_fro": [float(np.linalg.norm(st["st"][t])) for t in range(N_PTS)],
        "psit_fro": [float(np.linalg.norm(st["psit"][t])) for t in range(N_PTS)],
        "mt_norm": [float(np.linalg.norm(st["mt"][t])) for t in range(N_PTS)],
        "lamt_norm": [float(np.linalg.norm(st["lamt"][t])) for t in range(N_PTS)],
        "dEsde_ds_fro": [float(np.linalg.norm(st["dEsde_ds"][t])) for t in range(N_PTS)],
    }
    path = os.path.join(ROOT, "tests", "golden", "anchors_d4096.json" if len(sys.argv) <= 2 else f"anchors_d{D}_np{N_PTS}.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"oracle sweep {secs:.1f} s -> {path}")


if __name__ == "__main__":
    main()
