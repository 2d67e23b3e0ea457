#!/usr/bin/env python3
"""Summaries out of the SQLite files rocprofv3 (ROCm 7.2) writes by default.

    python tools/rocprof_db.py stats  <results.db> <out.csv>                   # per-kernel calls / total / avg / min / max (us)
    python tools/rocprof_db.py traffic <fetch.db> <write.db> B D Np <out.csv> <pmc_traffic.json>

`traffic`: per-launch HBM bytes of the sweep kernels from two counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-problem`.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM section): hbm_bytes = (2 FETCH + WRITE) * 1024,
the maximum over the dispatches of a kernel.  The correction is checked on the forward stepping kernel, whose reads are
exactly A and b."""
import collections
import json
import re
import sqlite3
import sys

# (first match wins: the backward kernel that assembles the gradient -- tenth template argument GF = true -- before the plain one)
KEYS = [(r"k_ode_(pe|sym)<\d+, true", "solve_fwd"), (r"k_ode_sym<\d+, false, \d+, \w+, -?\d+, \d+, \w+, \d+, \w+, true", "solve_bwd_grad"),
        (r"k_ode_(pe|sym)<\d+, false", "solve_bwd"), (r"k_energy_l96", "energy_l96"), (r"k_grad", "grad")]


def head_note():
    """The commit the measured tree was built from (tools/profile_*.sh export VGPA_HEAD: the GPU box has no .git)."""
    import os
    h = os.environ.get("VGPA_HEAD", "")
    return f"; tree {h}" if h else ""


def stats(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats (durations in us)" + head_note() + "\n")
        fh.write("Name,Calls,TotalDurationUs,AverageUs,Percentage,MinUs,MaxUs\n")
        for name, calls, tot, avg, mn, mx in rows:
            fh.write('"%s",%d,%.3f,%.3f,%.4f,%.3f,%.3f\n' % (name, calls, tot / 1e3, avg / 1e3, 100.0 * tot / total, mn / 1e3, mx / 1e3))
    for r in rows[:8]:
        print("%-70s calls %4d  avg %10.3f us" % (r[0][:70], r[1], r[3] / 1e3))


def per_dispatch(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    acc, name, grid = collections.defaultdict(float), {}, {}
    for did, kname, g, val in cur.execute("select dispatch_id, kernel_name, grid_size, value from counters_collection "
                                          "where counter_name = ?", (counter,)):
        acc[did] += float(val)
        name[did], grid[did] = kname, g
    out = collections.defaultdict(list)
    for d, v in acc.items():
        out[name[d]].append((v, grid[d]))
    return out


def traffic(fetch_db, write_db, B, D, Np, out_csv, out_json):
    B, D, Np = int(B), int(D), int(Np)
    f, w = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
    rows, res = [], {}
    for kname, vals in sorted(f.items()):
        short = next((v for k, v in KEYS if re.search(k, kname)), None)
        if short is None:
            continue
        fv, grid = max(vals)
        wv = max(w.get(kname, [(0.0, "")]))[0]
        hbm = (2.0 * fv + wv) * 1024.0
        rows.append((kname, grid, len(vals), fv, wv, hbm))
        res[f"{short}_B{B}_D{D}_Np{Np}"] = hbm
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 2 --warmup 1 "
                 f"--batch {B} --no-cpu-baseline --no-single-problem" + head_note() + "\n"
                 "# KB per dispatch (max over dispatches); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n")
        exp = B * Np * (D * D + D) * 8.0
        cal = [r for r in rows if re.search(KEYS[0][0], r[0])]
        if cal:
            fh.write(f"# calibration: forward stepping kernel reads A and b = {exp:.4e} B; 2*FETCH_SIZE*1024 = {2.0 * cal[0][3] * 1024.0:.4e} B\n")
        fh.write("kernel,grid_size,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected\n")
        for r in rows:
            fh.write('"%s",%s,%d,%.1f,%.1f,%.0f\n' % r)
    try:                                   # keep the entries of other batch sizes / dimensions
        old = json.load(open(out_json))
    except (OSError, ValueError):
        old = {}
    old.update(res)
    json.dump(old, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


def traffic_all(fetch_db, write_db, out_csv, note="", out_json=None, key=None, pattern=None):
    """Every kernel of a run: FETCH_SIZE / WRITE_SIZE per dispatch (max over dispatches), corrected bytes.  With out_json / key /
    pattern: the sum over the kernels whose name matches `pattern` is stored under `key` (bench.py reads it as `traffic`)."""
    f, w = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)" + (": " + note if note else "") + head_note() + "\n"
                 "# KB per dispatch (max over dispatches); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B "
                 "requests at 64 B, MI355X_MICROARCH.md HBM section)\n")
        fh.write("kernel,grid_size,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected\n")
        for kname, vals in sorted(f.items()):
            fv, grid = max(vals)
            wv = max(w.get(kname, [(0.0, "")]))[0]
            fh.write('"%s",%s,%d,%.1f,%.1f,%.0f\n' % (kname, grid, len(vals), fv, wv, (2.0 * fv + wv) * 1024.0))
            print("%-80s fetch %12.1f KB  write %12.1f KB  -> %.4e B" % (kname[:80], fv, wv, (2.0 * fv + wv) * 1024.0))
    if out_json and key and pattern:
        tot = 0.0
        for kname, vals in f.items():
            if re.search(pattern, kname):
                tot += (2.0 * max(vals)[0] + max(w.get(kname, [(0.0, "")]))[0]) * 1024.0
        try:
            old = json.load(open(out_json))
        except (OSError, ValueError):
            old = {}
        old[key] = tot
        json.dump(old, open(out_json, "w"), indent=1, sort_keys=True)
        print(key, "=", tot)


def counters(db_path, out_csv, note=""):
    """Every counter of a --pmc pass, per kernel: the sum over the dispatches' values (all XCDs / SEs / instances) averaged
    over the dispatches of the kernel, and -- where both were collected -- the matrix pipe's busy share
    SQ_VALU_MFMA_BUSY_CYCLES / (32 SQ_BUSY_CYCLES).  SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs of the chip, SQ_BUSY_CYCLES
    over its 32 shader engines (8 XCDs x 4: each counts the cycles ANY of its CUs is busy = the kernel's duration while it fills the
    chip), so mfma / (32 busy) = the share of SIMD-cycles the matrix pipe is busy (checked on the D = 40 forward stepper:
    9.175e9 / (32 x 5.09e8) = 0.56 against 0.54 from the rocprofv3 duration, profiles/r03c_sq_counters_B512.txt)."""
    cur = sqlite3.connect(db_path).cursor()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for did, kname, cname, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        acc[kname][cname] += float(val)
        disp[kname].add(did)
    names = sorted({c for k in acc for c in acc[k]})
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --pmc " + " ".join(names) + (": " + note if note else "") + head_note() + "\n")
        fh.write("# per kernel: counter sums over all instances, averaged over the kernel's dispatches\n")
        fh.write("kernel,dispatches," + ",".join(names) + ",mfma_busy_share_of_simd_cycles\n")
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0.0)):
            n = max(len(disp[k]), 1)
            vals = [acc[k].get(c, 0.0) / n for c in names]
            busy, mf = acc[k].get("SQ_BUSY_CYCLES", 0.0), acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            ratio = (mf / (32.0 * busy)) if busy > 0 else float("nan")
            fh.write('"%s",%d,%s,%.4f\n' % (k, n, ",".join("%.0f" % v for v in vals), ratio))
            if busy > 0 and mf > 0:
                print("%-70s dispatches %5d  matrix pipe busy %.3f of the SIMD-cycles" % (k[:70], n, ratio))


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(*sys.argv[2:5])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic_all":
        traffic_all(*sys.argv[2:9])
    else:
        traffic(*sys.argv[2:9])
