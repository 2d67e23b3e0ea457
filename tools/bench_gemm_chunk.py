#!/usr/bin/env python3
"""Time of one rank's stage product of BASELINE configs[4] (row block 512 x 4096, K = 4096) as the pipelined schedule launches it:
`chunks` K-chunk launches (vgpa_ld_gemm_chunk, segmented k-tiles), against ONE plain launch (vgpa_ld_gemm).  Prints one JSON line.
    python tools/bench_gemm_chunk.py [M] [D] [world] [chunks]        (VGPA_GEMM_PF=0: the two-register-set loop)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from legacy_sharded import HipStageBackend
    m, d, world, chunks = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 512), (2, 4096), (3, 8), (4, 4)))
    be = HipStageBackend()
    lib, st = be._lib, be._stream()
    dev = torch.device("cuda", 0)
    out = {"M": m, "D": d, "world": world, "chunks": chunks, "gemm_pf": os.environ.get("VGPA_GEMM_PF", "default")}
    for transa in (0, 1):
        a_d = torch.randn((d, m) if transa else (m, d), dtype=torch.float64, device=dev)
        x_d = torch.randn((d, d), dtype=torch.float64, device=dev)
        w_d = torch.zeros((m * d,), dtype=torch.float64, device=dev)
        mp = d // world
        sub = mp // chunks
        lda = m if transa else d

        def chunked():
            for j in range(chunks):
                a_off = j * sub * lda if transa else j * sub
                assert lib.vgpa_ld_gemm_chunk(st, transa, m, d, world * sub, be._p(a_d, a_off), lda, be._p(x_d, j * sub * d), d, be._p(w_d), d,
                                              sub // 16, mp, int(j > 0)) == 0

        def plain():
            assert lib.vgpa_ld_gemm(st, transa, m, d, d, be._p(a_d), None, lda, be._p(x_d), d, be._p(w_d), d) == 0

        for name, fn in (("chunked", chunked), ("plain", plain)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            be.sync() if hasattr(be, "sync") else torch.cuda.synchronize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[("TN" if transa else "NN") + "_" + name + "_us"] = round(1e6 * dt, 1)
            out[("TN" if transa else "NN") + "_" + name + "_tflops"] = round(2.0 * m * d * d / dt / 1e12, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
