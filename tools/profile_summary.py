#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (tools/profile_round.sh) into small files for profiles/:
  <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of bench.py (per-kernel average durations)
  <tag>_configs_kernel_stats.csv  ... of tools/bench_configs.py
  <tag>_traffic.json          HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes), corrected as
                              MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads;
                              unit KB), next to the algorithmic bytes
  <tag>_sq_c4.txt             SQ instruction counters of config 4
and refreshes profiles/traffic_latest.json for bench.py (with the kernel-source digest it applies to)."""
import csv, glob, json, os, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def first(pattern):
    g = glob.glob(os.path.join(O, pattern), recursive=True)
    return g[0] if g else None


def copy_stats(src_dir, dst):
    f = first(src_dir + "/**/*kernel_stats.csv")
    if f:
        open(os.path.join(P, dst), "w").write(open(f).read())
        return f
    return None


def pmc_per_kernel(dirname):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


copy_stats("kt_bench", tag + "_kernel_stats.csv")
copy_stats("kt_configs", tag + "_configs_kernel_stats.csv")

traffic = {"tag": tag, "correction": "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide "
           "coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE exact; both in KB; separate --pmc passes", "kernels": {}}
for which in ("bench", "cfg"):
    fs = pmc_per_kernel("pmc_%s_FETCH_SIZE" % which)
    ws = pmc_per_kernel("pmc_%s_WRITE_SIZE" % which)
    for k in fs:
        if not k.startswith("void smolhip") and "smolhip" not in k:
            continue
        f = fs[k].get("FETCH_SIZE", [])
        w = ws.get(k, {}).get("WRITE_SIZE", [])
        if not f or not w:
            continue
        # skip warm-up dispatches: take the median
        fmed = sorted(f)[len(f) // 2] * 1024.0 * 2.0
        wmed = sorted(w)[len(w) // 2] * 1024.0
        traffic["kernels"]["%s:%s" % (which, k[:90])] = {
            "dispatches": len(f), "hbm_read_bytes_per_launch": fmed, "hbm_write_bytes_per_launch": wmed,
            "hbm_bytes_per_launch": fmed + wmed}

try:
    line = json.loads(open(os.path.join(O, "bench_line.json")).read().strip().splitlines()[-1])
    algo = line["roofline"]["bytes_per_launch"]
    traffic["bench_line"] = {"launch_us": line["roofline"]["launch_us"], "frac": line["roofline"]["frac"],
                             "algorithmic_bytes_per_launch": algo}
    import bench
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    for k, v in traffic["kernels"].items():
        if k.startswith("bench:") and "k_pow2" in k:
            v["ratio_traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / algo
            latest = dict(v)
            latest.update({"kernel": k, "algorithmic_bytes_per_launch": algo, "source": "profiles/%s_traffic.json" % tag,
                           "commit": commit, "kernel_source_digest": bench.kernel_source_digest(),
                           "correction": traffic["correction"]})
            json.dump(latest, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
except Exception as e:      # noqa: BLE001
    traffic["bench_line_error"] = repr(e)

# algorithmic bytes of the side configs
try:
    for ln in open(os.path.join(O, "configs.jsonl")):
        d = json.loads(ln)
        traffic.setdefault("configs", {})[d["config"]] = {"kernel": d["kernel"], "step_us": d["step_us"],
                                                          "algorithmic_bytes_per_step": d["algorithmic_bytes_per_step"],
                                                          "frac_of_8TBps": d["frac_of_8TBps"]}
except Exception:           # noqa: BLE001
    pass
json.dump(traffic, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)

sq = pmc_per_kernel("sq_c4")
with open(os.path.join(P, tag + "_sq_c4.txt"), "w") as fh:
    fh.write("# SQ counters, config 4 (65535x4096 RGB8 -> 8192x4096 RGBA8), sums over the dispatches of 5 launches (tools/profile_round.sh)\n")
    for k, v in sq.items():
        if "smolhip" not in k:
            continue
        fh.write(k[:100] + "\n")
        for c, x in sorted(v.items()):
            fh.write("   %-28s n=%d  per launch %.6g\n" % (c, len(x), sum(x) / len(x)))
for f in ("configs.jsonl", "configs_opaque.jsonl", "bench_line.json"):
    src = os.path.join(O, f)
    if os.path.exists(src):
        open(os.path.join(P, "%s_%s" % (tag, f)), "w").write(open(src).read())
print("profiles written for", tag)
