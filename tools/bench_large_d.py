#!/usr/bin/env python3
"""Times the large-D (per-stage GEMM) forward and backward recursions on one GPU through the C ABI
(vgpa_solve_fwd / vgpa_solve_bwd on device-resident inputs are not exposed yet, so this times the kernels via the
context's profile events is not possible either: we time the host entry points and subtract nothing -- the
H2D/D2H copies of the (Np, D, D) arrays are included; see `kernel_only` for the stage kernels timed alone)."""
import os
import sys
import json
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import ctypes
    from vgpa_amd._lib import load
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    lib = load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(2)
    a = 8.0 * torch.eye(d, dtype=torch.float64, device=dev) + 0.05 * torch.randn(d, d, dtype=torch.float64, device=dev, generator=g) / np.sqrt(d)
    x = torch.randn(d, d, dtype=torch.float64, device=dev, generator=g)
    x = x + x.T
    c = torch.zeros(d, d, dtype=torch.float64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {"D": d}
    for transa in (0, 1):
        for _ in range(3):
            lib.vgpa_ld_gemm(st, transa, d, d, d, p(a), None, d, p(x), d, p(c), d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.vgpa_ld_gemm(st, transa, d, d, d, p(a), None, d, p(x), d, p(c), d)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ref = (a.T if transa else a) @ x
        err = float((c - ref).abs().max() / ref.abs().max())
        out["gemm_T" if transa else "gemm_N"] = {"ms": ms, "tflops": 2.0 * d ** 3 / ms / 1e9, "rel_err_vs_torch": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
