#!/usr/bin/env python3
"""Times the stage product of ONE rank of the row-sharded recursion (BASELINE configs[4]: D = 4096 over 8 ranks -> a 512 x 4096
x 4096 fp64 GEMM, packed output) on one GPU, whole and split into K-chunks / N-chunks, next to the full 4096^3 product.
    python tools/bench_shard_gemm.py [--dim 4096] [--world 8] [--reps 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from legacy_sharded import HipStageBackend
    be = HipStageBackend()
    lib = be._lib
    d, mp = args.dim, args.dim // args.world
    dev = torch.device("cuda", 0)
    a = torch.randn((d, d), dtype=torch.float64, device=dev)
    x = torch.randn((d, d), dtype=torch.float64, device=dev)
    w = torch.zeros((d, d), dtype=torch.float64, device=dev)
    st = be._stream()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        for _ in range(args.reps):
            fn()
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    out = {"D": d, "world": args.world, "Mp": mp}
    for transa in (0, 1):
        tag = "TN" if transa else "NN"
        a_ptr = be._p(a) if not transa else be._p(a)        # rows I_0 of A / columns I_0 of A
        full = timed(lambda: lib.vgpa_ld_gemm(st, transa, d, d, d, be._p(a), None, d, be._p(x), d, be._p(w), d))
        blk = timed(lambda: lib.vgpa_ld_gemm(st, transa, mp, d, d, a_ptr, None, d, be._p(x), d, be._p(w), mp))
        out[tag] = {"full_ms": full, "full_tflops": 2.0 * d ** 3 / full / 1e9, "block_ms": blk,
                    "block_tflops": 2.0 * mp * d * d / blk / 1e9, "strong_scaling_gemm_only": full / blk}
        for c in (2, 4, 8):
            # N-chunks: C launches of Mp x (D/C) x D
            nc = d // c
            t = timed(lambda: [lib.vgpa_ld_gemm(st, transa, mp, nc, d, a_ptr, None, d, be._p(x, j * nc), d, be._p(w, j * mp * nc), nc)
                               for j in range(c)])
            out[tag][f"n_chunks_{c}_ms"] = t
            # K-chunks: C launches of Mp x D x (D/C) (timing only: the shipped kernel has no accumulate yet)
            kc = d // c
            t = timed(lambda: [lib.vgpa_ld_gemm(st, transa, mp, d, kc, be._p(a, (j * kc * d) if transa else j * kc), None, d,
                                                be._p(x, j * kc * d), d, be._p(w), mp) for j in range(c)])
            out[tag][f"k_chunks_{c}_ms"] = t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
