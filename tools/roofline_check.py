#!/usr/bin/env python3
"""Re-derives every roofline fraction of the bench line from the files TRACKED under profiles/ alone:

    frac = algorithmic bytes per launch / mean kernel duration / 8 TB/s

with the duration taken from the rocprofv3 kernel trace of that workload's OWN run (the last timed dispatches of its
(kernel, grid) group: profiles/<tag>_kernel_by_grid.csv; the headline alone in one run, one run per side config, so no
average mixes two workloads or the clock-settling launches). Per row, three figures side by side:

  * trace   the fraction re-derived from the kernel trace (kernel time only);
  * events  what the SAME traced process measured with hipEvents (profiles/<tag>_traced_runs.jsonl; includes the
            dispatch gaps, which the profiler widens for by-value single-image launches);
  * line    the untraced default run of bench.py (profiles/<tag>_bench_line.json), a different process: two processes
            on one box differ by up to +-5 % (clock and thermal state).

Exit status 1 when a trace-derived fraction differs from the line's by more than --tol (default 8 %: the method error seen against the same-process events column is 1-2 %, the rest is the
run-to-run spread).

    python tools/roofline_check.py r03final [--tol 0.08]
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
ap.add_argument("--tol", type=float, default=0.08)
args = ap.parse_args()
P = os.path.join(ROOT, "profiles")
line = json.loads(open(os.path.join(P, args.tag + "_bench_line.json")).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(os.path.join(P, args.tag + "_kernel_by_grid.csv"))))
traced = {}
for ln in open(os.path.join(P, args.tag + "_traced_runs.jsonl")):
    d = json.loads(ln)
    traced[d["run"]] = d


def run_time_us(run):
    """mean duration of one step of `run': the dominant (kernel, grid) group, or the sum over the groups that were each
    dispatched about as often as the dominant one (multi-kernel steps: k_boxh + k_boxv)"""
    g = [r for r in rows if r["run"] == run]
    if not g:
        return None, []
    top = max(int(r["dispatches"]) for r in g)
    used = [r for r in g if int(r["dispatches"]) * 2 > top]
    return sum(float(r["mean_us"]) for r in used), used


bad = 0
print("%-8s %-40s %12s | %9s %7s | %9s %7s | %9s %7s %7s" % ("row", "kernel (grid)", "algo bytes", "trace us", "frac",
                                                          "events us", "frac", "line us", "frac", "spread"))


def report(name, run, algo, line_us, line_frac):
    global bad
    us, used = run_time_us(run)
    if us is None:
        print("%-8s no kernel trace for run %s" % (name, run))
        bad += 1
        return
    frac = algo / (us * 1e-6) / PEAK
    what = " + ".join("%s (%s)" % (r["kernel"].split("(")[0].replace("void smolhip::", ""), r["grid"]) for r in used)
    t = traced.get(run)
    if t is None:
        ev_us = ev_frac = float("nan")
    elif run == "bench_c2":
        ev_us, ev_frac = t["roofline"]["launch_us"], t["roofline"]["frac"]
    else:
        # (tools/bench_configs.py draws C5mix's pixel types from its own seed: its bytes per step differ slightly)
        ev_us, ev_frac = t["step_us"], t["frac_of_8TBps"]
        frac = t["algorithmic_bytes_per_step"] / (us * 1e-6) / PEAK
    ok = abs(frac / line_frac - 1) <= args.tol
    print("%-8s %-40s %12d | %9.2f %7.4f | %9.2f %7.4f | %9.2f %7.4f %+6.1f%% %s"
          % (name, what[:40], algo, us, frac, ev_us, ev_frac, line_us, line_frac, (frac / line_frac - 1) * 100,
             "" if ok else "<-- differs from the line"))
    bad += 0 if ok else 1


r = line["roofline"]
report("C2", "bench_c2", r["bytes_per_launch"], r["launch_us"], r["frac"])
for name, d in line.get("other_baseline_configs", {}).items():
    if "error" in d:
        continue
    key = name.split(" ", 1)[0]
    report(key, {"C2CROP": "C2CROP4"}.get(key, key), d["algorithmic_bytes_per_step"], d["step_us"], d["frac_of_peak"])
print("roofline_check: %s" % ("OK" if not bad else "%d row(s) differ" % bad))
sys.exit(1 if bad else 0)
