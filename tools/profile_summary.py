#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (tools/profile_round.sh) into small files for profiles/:
  <tag>_kernel_by_grid.csv    per (run, kernel symbol, grid, workgroup): number of dispatches, and mean / min / median
                              duration of the LAST n dispatches (n = the run's timed steps: the clock-settling and
                              warm-up launches in front of them are dropped) -- what tools/roofline_check.py reads
  <tag>_kernel_stats.csv      rocprofv3's own --stats table of the headline-only bench.py run
  <tag>_traffic.json          HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes), corrected as
                              MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads;
                              unit KB), next to the algorithmic bytes, headline and side configs
  <tag>_sq_box.txt            SQ instruction / LDS counters of the BOX row
and refreshes profiles/traffic_latest.json + traffic_configs_latest.json for bench.py (with the kernel-source digest
they apply to)."""
import collections
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def first(pattern):
    g = glob.glob(os.path.join(O, pattern), recursive=True)
    return g[0] if g else None


def trace_rows(run_dir):
    """[(kernel, grid string, workgroup string, duration us)] in dispatch order"""
    f = first(run_dir + "/**/*kernel_trace.csv")
    rows = []
    if not f:
        return rows
    for r in csv.DictReader(open(f)):
        if "smolhip" not in r["Kernel_Name"] or "k_zone_" in r["Kernel_Name"]:      # (k_zone_read2: smol_hip_memory_zones' probe, no part of a step)
            continue
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"],
                     "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]),
                     "%sx%sx%s" % (r["Workgroup_Size_X"], r["Workgroup_Size_Y"], r["Workgroup_Size_Z"]),
                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    return [r[1:] for r in rows]


def by_grid(run, run_dir, last_n, out):
    groups = collections.OrderedDict()
    for k, g, wg, us in trace_rows(run_dir):
        groups.setdefault((k, g, wg), []).append(us)
    for (k, g, wg), us in groups.items():
        tail = us[-last_n:] if last_n and len(us) > last_n else us
        out.append({"run": run, "kernel": k, "grid": g, "workgroup": wg, "dispatches": len(us), "timed": len(tail),
                    "mean_us": round(sum(tail) / len(tail), 3), "min_us": round(min(tail), 3),
                    "median_us": round(statistics.median(tail), 3),
                    "mean_all_us": round(sum(us) / len(us), 3)})        # every dispatch, clock-settling launches included (= rocprofv3 --stats)


rows = []
by_grid("bench_c2", "kt_bench", 20, rows)                      # bench.py --steps 20: the last 20 launches of each (kernel, grid)
for d in sorted(glob.glob(os.path.join(O, "kt_cfg_*"))):
    if os.path.isdir(d):
        c = os.path.basename(d)[len("kt_cfg_"):]
        # tools/bench_configs.py --iters 20: settle + warm-up, `iters' launches in one timed region, then 20 single steps
        by_grid(c, os.path.basename(d), 40, rows)
with open(os.path.join(P, tag + "_kernel_by_grid.csv"), "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=["run", "kernel", "grid", "workgroup", "dispatches", "timed", "mean_us", "min_us", "median_us", "mean_all_us"])
    wr.writeheader()
    wr.writerows(rows)

# what the traced runs measured themselves (hipEvents inside the traced process): the line of the traced bench.py run
# and the tools/bench_configs.py line of every traced side config -- tools/roofline_check.py compares trace and events
# of the SAME process (run-to-run spread on one box is +-5 %, so a second process is no yardstick)
with open(os.path.join(P, tag + "_traced_runs.jsonl"), "w") as fh:
    for log, run in [("kt_bench.log", "bench_c2")] + [(os.path.basename(d) + ".log", os.path.basename(d)[len("kt_cfg_"):])
                                                       for d in sorted(glob.glob(os.path.join(O, "kt_cfg_*"))) if os.path.isdir(d)]:
        try:
            lines = [ln for ln in open(os.path.join(O, log)).read().splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1])
            d["run"] = run
            fh.write(json.dumps(d) + "\n")
        except Exception:       # noqa: BLE001
            pass

f = first("kt_bench/**/*kernel_stats.csv")
if f:
    open(os.path.join(P, tag + "_kernel_stats.csv"), "w").write(open(f).read())


def pmc_per_kernel(dirname):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


CORRECTION = ("MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced "
              "(16 B/lane) streaming read -> doubled; WRITE_SIZE exact; both in KB; separate --pmc passes; median over "
              "the dispatches of a kernel symbol (warm-up included), summed over the symbols of a multi-kernel step")
traffic = {"tag": tag, "correction": CORRECTION, "kernels": {}, "configs": {}}


def hbm_bytes(fetch_dir, write_dir):
    """({kernel: bytes per launch}, total) -- the kernels of one run, each by its median dispatch"""
    fs, ws = pmc_per_kernel(fetch_dir), pmc_per_kernel(write_dir)
    per = {}
    for k in fs:
        if "smolhip" not in k or "k_zone_" in k:
            continue
        f = fs[k].get("FETCH_SIZE", [])
        w = ws.get(k, {}).get("WRITE_SIZE", [])
        if not f or not w:
            continue
        per[k[:100]] = {"dispatches": len(f), "hbm_read_bytes_per_launch": statistics.median(f) * 1024.0 * 2.0,
                        "hbm_write_bytes_per_launch": statistics.median(w) * 1024.0}
        per[k[:100]]["hbm_bytes_per_launch"] = per[k[:100]]["hbm_read_bytes_per_launch"] + per[k[:100]]["hbm_write_bytes_per_launch"]
    return per


import bench  # noqa: E402
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()

per = hbm_bytes("pmc_bench_FETCH_SIZE", "pmc_bench_WRITE_SIZE")
for k, v in per.items():
    traffic["kernels"]["bench:" + k] = v
try:
    line = json.loads(open(os.path.join(O, "bench_line.json")).read().strip().splitlines()[-1])
    algo = line["roofline"]["bytes_per_launch"]
    traffic["bench_line"] = {"launch_us": line["roofline"]["launch_us"], "frac": line["roofline"]["frac"],
                             "algorithmic_bytes_per_launch": algo}
    for k, v in per.items():
        if "k_pow2" in k:
            v["ratio_traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / algo
            latest = dict(v)
            prof = [r for r in rows if r["run"] == "bench_c2" and "k_pow2" in r["kernel"]]
            if prof:
                latest["profiled_launch_us"] = prof[0]["mean_us"]
                latest["profiled_launch_all_us"] = prof[0]["mean_all_us"]
                latest["profiled_source"] = ("profiles/%s_kernel_by_grid.csv: mean of the last %d of %d dispatches, rocprofv3 "
                                             "--kernel-trace of bench.py alone" % (tag, prof[0]["timed"], prof[0]["dispatches"]))
            latest.update({"kernel": k, "algorithmic_bytes_per_launch": algo, "source": "profiles/%s_traffic.json" % tag,
                           "commit": commit, "kernel_source_digest": bench.kernel_source_digest(), "correction": CORRECTION})
            json.dump(latest, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
except Exception as e:      # noqa: BLE001
    traffic["bench_line_error"] = repr(e)

algo_of = {}
try:
    for ln in open(os.path.join(O, "configs.jsonl")):
        d = json.loads(ln)
        algo_of[d["config"]] = d
except Exception:           # noqa: BLE001
    pass
cfg_latest = {"kernel_source_digest_all": bench.kernel_source_digest(all_kernels=True), "commit": commit,
              "correction": CORRECTION, "configs": {}}
for d in sorted(glob.glob(os.path.join(O, "pmc_cfg_*_FETCH_SIZE"))):
    c = os.path.basename(d)[len("pmc_cfg_"):-len("_FETCH_SIZE")]
    per = hbm_bytes(os.path.basename(d), "pmc_cfg_%s_WRITE_SIZE" % c)
    if not per:
        continue
    total = sum(v["hbm_bytes_per_launch"] for v in per.values())
    entry = {"kernels": per, "hbm_bytes_per_launch": total, "source": "profiles/%s_traffic.json" % tag}
    if c in algo_of:
        entry["algorithmic_bytes_per_step"] = algo_of[c]["algorithmic_bytes_per_step"]
        entry["ratio_traffic_over_algorithmic"] = total / algo_of[c]["algorithmic_bytes_per_step"]
        entry["step_us"] = algo_of[c]["step_us"]
        entry["frac_of_8TBps"] = algo_of[c]["frac_of_8TBps"]
    traffic["configs"][c] = entry
    # bench.py's side table names its rows "C1 ...", "BOX ...", "C2CROP ..." (first word)
    cfg_latest["configs"]["C2CROP" if c == "C2CROP4" else c] = entry
json.dump(traffic, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)
json.dump(cfg_latest, open(os.path.join(P, "traffic_configs_latest.json"), "w"), indent=1)

sq = pmc_per_kernel("sq_box")
with open(os.path.join(P, tag + "_sq_box.txt"), "w") as fh:
    fh.write("# SQ counters, BOX row (4096^2 -> 448^2 RGBA8 premultiplied, 8 frames per launch), mean per launch over the dispatches "
             "of each pass (tools/profile_round.sh)\n")
    for k, v in sq.items():
        if "smolhip" not in k or "k_zone_" in k:
            continue
        fh.write(k[:110] + "\n")
        for c, x in sorted(v.items()):
            fh.write("   %-28s n=%d  per launch %.6g\n" % (c, len(x), sum(x) / len(x)))
        bc, ia = v.get("SQ_LDS_BANK_CONFLICT"), v.get("SQ_LDS_IDX_ACTIVE")
        if bc and ia:
            fh.write("   LDS bank-conflict cycles / LDS active cycles = %.3f\n" % ((sum(bc) / len(bc)) / (sum(ia) / len(ia))))
for c in ("SRGB", "SRGBUP", "SRGBHALF", "SRGBPHOTO", "SRGBBOX"):
    src = os.path.join(O, "sq_%s.txt" % c)
    if os.path.exists(src):
        open(os.path.join(P, "%s_sq_%s.txt" % (tag, c.lower())), "w").write(open(src).read())
for f in ("configs.jsonl", "configs_opaque.jsonl", "bench_line.json"):
    src = os.path.join(O, f)
    if os.path.exists(src):
        open(os.path.join(P, "%s_%s" % (tag, f)), "w").write(open(src).read())
print("profiles written for", tag)
