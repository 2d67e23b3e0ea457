#!/usr/bin/env python3
"""Timeline of ONE batch of the D > 64 energy terms (lde_energy) out of a rocprofv3 --kernel-trace database:
    python tools/trace_lde.py <results.db>
Takes the last k_prep ... SYRK (k_gemm_bv<true> or k_gemm_b<true, ...>) window and prints, per phase, the wall time between the
first start and the last end of its dispatches, the sum of their durations and their number (the look-ahead of the Cholesky phase
shows as wall < sum)."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
lde = [(n, s, e) for n, s, e in rows if "lde::" in n]
last_prep = max(i for i, r in enumerate(lde) if "k_prep" in r[0])
win = lde[last_prep:]
stop = next(i for i, r in enumerate(win) if "k_gemm_bv<true" in r[0] or "k_gemm_b<true" in r[0])
win = win[:stop + 1]
zu = next(i for i, r in enumerate(win) if "k_zero_upper" in r[0])
mv = next(i for i, r in enumerate(win) if "k_matvec" in r[0])
phases = [("prep", win[:1]), ("cholesky (diag, panel, trailing)", win[1:zu + 1]), ("inverse", win[zu + 1:mv - 1]), ("G = A L", win[mv - 1:mv]),
          ("matvec, residuals, scalars, scale", win[mv:-1]), ("SYRK", win[-1:])]
t0 = win[0][1]
print("columns of `kernels`:", cols)
print("%-40s %6s %10s %10s" % ("phase", "calls", "wall us", "sum us"))
for name, ds in phases:
    if not ds:
        continue
    wall = (max(d[2] for d in ds) - min(d[1] for d in ds)) / 1e3
    print("%-40s %6d %10.1f %10.1f" % (name, len(ds), wall, sum(d[2] - d[1] for d in ds) / 1e3))
print("%-40s %6d %10.1f" % ("batch", len(win), (win[-1][2] - t0) / 1e3))
by = {}
for n, s, e in win[1:zu + 1]:
    k = n.split("(")[0][-40:]
    by.setdefault(k, []).append((e - s) / 1e3)
for k, v in by.items():
    print("   cholesky: %-42s calls %3d  sum %8.1f  avg %7.1f" % (k, len(v), sum(v), sum(v) / len(v)))
if "-v" in sys.argv:
    for n, s, e in win:
        print("%10.1f %10.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
