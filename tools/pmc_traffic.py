"""Turns two rocprofv3 counter-collection CSVs (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, both of
`python bench.py --steps 2 --warmup 1 --batch B --no-cpu-baseline`) into per-launch HBM bytes of the sweep kernels.

    python tools/pmc_traffic.py fetch.csv write.csv B D Np profiles/<name>.csv profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM
section), so hbm_bytes = (2*FETCH + WRITE) * 1024.  The correction is checked on the forward stepping kernel, whose reads
are exactly A and b.  A dispatch's counter is the sum of the rows rocprofv3 emits for it; the figure kept per kernel is
the maximum over the batched dispatches (the single-problem launches of bench.py's latency probe are much smaller)."""
import collections
import csv
import json
import sys

KEYS = {"k_fwd_mfma": "solve_fwd", "k_bwd_mfma": "solve_bwd", "k_energy_l96": "energy_l96", "k_grad": "grad"}


def per_dispatch(path, counter):
    acc = collections.defaultdict(float)
    name = {}
    grid = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
        grid[r["Dispatch_Id"]] = r["Grid_Size"]
    out = collections.defaultdict(list)
    for d, v in acc.items():
        out[name[d]].append((v, grid[d]))
    return out


def main():
    fetch, write, B, D, Np, out_csv, out_json = sys.argv[1:8]
    B, D, Np = int(B), int(D), int(Np)
    f, w = per_dispatch(fetch, "FETCH_SIZE"), per_dispatch(write, "WRITE_SIZE")
    rows, res = [], {}
    for kname, vals in sorted(f.items()):
        short = next((v for k, v in KEYS.items() if k in kname), None)
        if short is None:
            continue
        fv, grid = max(vals)
        wv = max(w.get(kname, [(0.0, "")]))[0]
        hbm = (2.0 * fv + wv) * 1024.0
        rows.append((kname, grid, len(vals), fv, wv, hbm))
        res[f"{short}_B{B}_D{D}_Np{Np}"] = hbm
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 2 --warmup 1 "
                 f"--batch {B} --no-cpu-baseline\n# KB per dispatch (max over dispatches); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n")
        exp = B * Np * (D * D + D) * 8.0
        got = res.get(f"solve_fwd_B{B}_D{D}_Np{Np}")
        if got is not None:
            rd = 2.0 * next(r[3] for r in rows if "k_fwd_mfma" in r[0]) * 1024.0
            fh.write(f"# calibration: forward stepping kernel reads A and b = {exp:.4e} B; 2*FETCH_SIZE*1024 = {rd:.4e} B\n")
        fh.write("kernel,grid_size,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected\n")
        for r in rows:
            fh.write('"%s",%s,%d,%.1f,%.1f,%.0f\n' % r)
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
