#!/usr/bin/env python3
"""Re-derives every roofline fraction of the bench line from the files TRACKED under profiles/ alone and compares:

    frac = algorithmic bytes per launch / mean kernel duration (rocprofv3 kernel trace, last timed dispatches of that
           workload's own run: profiles/<tag>_kernel_by_grid.csv) / 8 TB/s

for the headline (config 2) and for every row of the line's side table, against what bench.py measured with hipEvents
(profiles/<tag>_bench_line.json). Exit status 1 when a fraction differs by more than --tol (default 3 %, plus the
dispatch gaps hipEvents see and a kernel trace does not: 1.5 us per launch).

    python tools/roofline_check.py r03final [--tol 0.03]
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0e9

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--tol", type=float, default=0.03)
args = ap.parse_args()
P = os.path.join(ROOT, "profiles")
line = json.loads(open(os.path.join(P, args.tag + "_bench_line.json")).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(os.path.join(P, args.tag + "_kernel_by_grid.csv"))))


def run_time_us(run):
    """mean duration of one step of `run': the dominant (kernel, grid) group, or the sum over the groups that were each
    dispatched about as often as the dominant one (multi-kernel steps: k_boxh + k_boxv, mixed batches)"""
    g = [r for r in rows if r["run"] == run]
    if not g:
        return None, []
    top = max(int(r["dispatches"]) for r in g)
    used = [r for r in g if int(r["dispatches"]) * 2 > top]
    return sum(float(r["mean_us"]) for r in used), used


bad = 0
print("%-10s %-38s %12s %10s %10s %8s %8s" % ("row", "kernel (grid)", "algo bytes", "trace us", "bench us", "trace f", "bench f"))


def report(name, run, algo, bench_us, bench_frac):
    global bad
    us, used = run_time_us(run)
    if us is None:
        print("%-10s no kernel trace for run %s" % (name, run))
        bad += 1
        return
    frac = algo / (us * 1e-6) / PEAK
    what = " + ".join("%s (%s)" % (r["kernel"].split("(")[0].replace("void smolhip::", ""), r["grid"]) for r in used)
    ok = abs(frac - bench_frac) <= args.tol * bench_frac + 1.5 / max(us, 1e-9) * bench_frac
    print("%-10s %-38s %12d %10.2f %10.2f %8.4f %8.4f %s" % (name, what[:38], algo, us, bench_us, frac, bench_frac, "" if ok else "<-- differs"))
    bad += 0 if ok else 1


r = line["roofline"]
report("C2", "bench_c2", r["bytes_per_launch"], r["launch_us"], r["frac"])
for name, d in line.get("other_baseline_configs", {}).items():
    if "error" in d:
        continue
    key = name.split(" ", 1)[0]
    run = {"C2CROP": "C2CROP4"}.get(key, key)
    # tools/bench_configs.py builds the same images per launch as bench.py's side table, so the bytes per step agree
    report(key, run, d["algorithmic_bytes_per_step"], d["step_us"], d["frac_of_peak"])
print("roofline_check: %s" % ("OK" if not bad else "%d row(s) differ" % bad))
sys.exit(1 if bad else 0)
