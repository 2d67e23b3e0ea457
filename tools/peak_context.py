#!/usr/bin/env python3
"""profiles/peak_context.json -- what `peak` of the fp64 matrix pipe is worth on this part, MEASURED: parsed from the outputs of
tools/clock_under_load.sh (sclk / package power sampled while bench.py loops) and tools/sustained_peak.sh (the pipe's sustained
rate with nothing else on the chip).  bench.py reports the file's contents as `peak_context` instead of literals.

    python tools/peak_context.py <clock_under_load.txt> <sustained_fp64_peak.txt> [out.json]
"""
import json
import re
import statistics
import sys


def main():
    clock_txt, peak_txt = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/peak_context.json"
    load, idle = open(clock_txt).read().split("idle:")[0], open(clock_txt).read().split("idle:")[-1]
    sclk = [int(v) for v in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", load)]
    power = [float(v) for v in re.findall(r"Package Power \(W\): ([\d.]+)", load)]
    best, best_line = 0.0, ""
    lines = open(peak_txt).read().splitlines()
    for i, ln in enumerate(lines):
        m = re.match(r"SUSTAINED (.*): ([\d.]+) TFLOP/s", ln)
        if m and "zero" not in m.group(1) and float(m.group(2)) > best:
            best, best_line = float(m.group(2)), ln.strip()
            ctx = [c for c in lines[max(0, i - 4):i]]
            pk_clk = [int(v) for c in ctx for v in re.findall(r"\((\d+)Mhz\)", c)]
            pk_w = [float(v) for c in ctx for v in re.findall(r"([\d.]+) W", c)]
    res = {"fp64_matrix_nominal_TFLOPs": 78.6,
           "fp64_matrix_sustained_TFLOPs": best, "sustained_how": best_line + " (tools/sustained_peak.sh, tools/ubench/f64_sustained.hip)",
           "sustained_clock_GHz": (statistics.median(pk_clk) / 1e3) if pk_clk else None,
           "sustained_package_W": statistics.median(pk_w) if pk_w else None,
           "clock_GHz_under_this_bench": (statistics.median(sclk) / 1e3) if sclk else None,
           "package_W_under_this_bench": statistics.median(power) if power else None,
           "samples_under_bench": len(sclk),
           "idle": " ".join(idle.split())[:200],
           "power_capped": bool(sclk and statistics.median(sclk) < 2300),
           "source_files": [clock_txt, peak_txt]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
