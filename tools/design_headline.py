#!/usr/bin/env python3
"""Rewrites the two headline paragraphs, the opaque-sources sentence and the config table of DESIGN.md section 4.1 (round 5) from the
evidence files under profiles/ (r05final_bench_line.json, _c4_line.json, _c5mix_line.json, _configs*.jsonl, traffic_latest.json), so
that the text always quotes the LAST evidence run.    python tools/design_headline.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(path):
    out = {}
    for l in open(path):
        try:
            d = json.loads(l)
        except ValueError:
            continue
        if "config" in d:
            out[d["config"]] = d
    return out


d = json.load(open(os.path.join(P, "r05final_bench_line.json")))
r, e, cb = d["roofline"], d["e2e"], d["cpu_baseline"]
c4 = json.load(open(os.path.join(P, "r05final_c4_line.json")))
c5 = json.load(open(os.path.join(P, "r05final_c5mix_line.json")))
g = c5["graph_mode"]
t = json.load(open(os.path.join(P, "traffic_latest.json")))
rp = e["rows_per_call"]
f = {"launch": r["launch_us"], "tbs": r["achieved"] / 1000.0, "frac": 100 * r["frac"], "trace": t["profiled_launch_us"],
     "tfrac": 100 * r["bytes_per_launch"] / (t["profiled_launch_us"] * 1e-6) / 8e12,
     "afrac": 100 * r["bytes_per_launch"] / (t["profiled_launch_all_us"] * 1e-6) / 8e12, "value": d["value"],
     "p1": e["pinned"]["one_caller"]["Mpix_out_per_s"], "p16": e["pinned"]["16_callers"]["Mpix_out_per_s"],
     "pg": e["pinned"]["one_caller"]["GBps_over_pcie"], "q1": e["pageable"]["one_caller"]["Mpix_out_per_s"],
     "q16": e["pageable"]["16_callers"]["Mpix_out_per_s"], "ref": cb["value"], "avx1": cb["avx2_1thread_Mpix_s"],
     "gen1": cb["generic_1thread_Mpix_s"], "lib": cb["library_cpu_backend"]["Mpix_s"], "libt": cb["library_cpu_backend"]["threads"],
     "gmed": g["replay_us"]["median"] / 1000.0, "gmin": g["replay_us"]["min"] / 1000.0, "gips": 10000 / (g["replay_us"]["min"] * 1e-6),
     "single": g["single_launch_us"]["median"] / 1000.0, "ratio": g["replay_over_single_launch"], "cap": g["capture_and_instantiate_ms"],
     "c4f": c4["roofline"]["frac"], "c4p": c4["e2e"]["pinned"]["GBps_over_pcie"], "c4q": c4["e2e"]["pageable"]["GBps_over_pcie"]}
for key, name in (("1080p_to_720p.pinned", "a"), ("1080p_to_720p.pageable", "b"), ("c2_4096_to_1024.pinned", "c"), ("c2_4096_to_1024.pageable", "e")):
    for n in ("1", "8", "64", "all"):
        f[name + n] = rp[key][n]["ms_per_image"]
        f[name + n + "x"] = rp[key][n]["x_whole"]

TEXT = """Headline (`bench.py`, C2, 32 frames per launch; `r05final_bench_line.json`, `r05final_roofline_check.txt`: every row within 8 % between the
line, the traced process and its events): launch {launch:.1f} µs in the untraced line = {tbs:.2f} TB/s = **{frac:.1f} %** of 8 TB/s; the traced process of the
same run {trace:.1f} µs by rocprofv3 over its last 20 dispatches ({tfrac:.1f} %), {afrac:.1f} % over all of them (`frac_profiled_all`). **The headline has two
modes per process**, as C5mix has: ten fresh `bench.py` processes on one box ran 337-338 µs (0.845) seven times and 325-326 µs (0.875) three
times, whatever their flags (`profiles/r05_headline_modes.txt`) -- a property of the process, not of the code and not of the arena's
addresses (twelve rebuilds of the arena inside one process, behind fillers of 0 MB-1 GB, all stayed in that process's mode:
`tools/exp_placement.py`) nor of the stream (sixteen streams of one process: one mode, `tools/exp_streams.py`); the round's evidence runs
saw both (lines 336.3-338.2 µs, traces 319.9-337.4; the default line in `r05final_bench_line_default.json`: 103 197 Mpix/s, 0.879). HBM
traffic from the PMC passes 2 281.8 MB = **1.0001 ×** algorithmic; {value:.0f} Mpix_out/s. The headline kernel got no work this round.
The same frames from HOST memory through `smol_scale_new/batch/destroy` (`e2e`; the mapped road of §3 (e)): page-locked **{p1:.0f}** (one caller) /
{p16:.0f} (16 callers) Mpix/s = {pg:.1f} GB/s over the link (807-847 by box; round 4: 745 / 747), pageable {q1:.0f} / {q16:.0f} in this harness (613-711 by
box and harness: `profiles/r05_mapped_road.txt`). Row-at-a-time callers (`e2e.rows_per_call`; ms per image, and × THIS round's whole-image
call, which itself fell from 0.250 to {aall:.3f} ms page-locked and from 0.42 to {ball:.2f} pageable): 1080p → 720p page-locked 1 row {a1:.2f} ms ({a1x:.1f} ×;
round 4: 11.3 ms), 8 rows {a8:.2f} ({a8x:.1f} ×), 64 rows **{a64:.3f}** ({a64x:.2f} ×; round 4: 0.605); pageable {b1:.2f} / {b8:.2f} / {b64:.2f} ({b1x:.1f} × / {b8x:.1f} × / {b64x:.1f} ×); C2
page-locked {c1:.1f} / {c8:.1f} / **{c64:.2f}** ms ({c1x:.1f} × / {c8x:.1f} × / {c64x:.2f} ×), pageable {e1:.1f} / {e8:.1f} / {e64:.2f} ({e1x:.1f} × / {e8x:.1f} × / {e64x:.2f} ×); under the resident-source
promise: §9 item 4. Reference AVX2 on the box's 256 threads: {ref:.0f} Mpix/s; AVX2 one thread {avx1:.0f}, generic one thread {gen1:.1f}; the library's
CPU backend on its default {libt} threads {lib:.0f}.

BASELINE's own two measurements: config 5 (mixed types) as ONE hipGraph of 10 000 per-image kernel nodes: replay **{gmed:.1f} ms** (median; {gmin:.1f}
at best = {gips:.0f} images/s) against **{single:.3f} ms** for the descriptor-table launch = {ratio:.0f} ×, capture + instantiate {cap:.1f} ms, outputs verified
(`r05final_c5mix_line.json` `graph_mode`); config 4 (65535×4096 RGB8 → 8192×4096) kernel-only {c4f:.3f} of the roofline, from host memory
{c4p:.1f} GB/s page-locked / {c4q:.1f} GB/s pageable over the link (`r05final_c4_line.json` `e2e`).

""".format(**f)

path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("Headline (`bench.py`, C2, 32 frames per launch; `r05final_bench_line.json`")
b = s.index("Every row below is bit-exact against the oracle in the same run")
s = s[:a] + TEXT + s[b:]
a = s.index("| config | what | kernel | step | % of 8 TB/s | round 4 (`r04final_configs.jsonl`) |")
b = s.index("\n\n", a)
rows_old = s[a:b].split("\n")
new_c, old_c = load(os.path.join(P, "r05final_configs.jsonl")), load(os.path.join(P, "r04final_configs.jsonl"))
lines = rows_old[:2]
for row in rows_old[2:]:
    cfg = row.split("|")[1].strip()
    dd = new_c.get(cfg)
    if not dd:
        lines.append(row)
        continue
    o = old_c.get(cfg)
    r4 = ("%.1f µs, %.1f %%" % (o["step_us"], 100 * o["frac_of_8TBps"])) if o else row.split("|")[6].strip()
    lines.append("| %s | %s | `%s` | %.1f µs | %.1f %% | %s |" % (cfg, dd["desc"], dd["kernel"], dd["step_us"], 100 * dd["frac_of_8TBps"], r4))
s = s[:a] + "\n".join(lines) + s[b:]
op = load(os.path.join(P, "r05final_configs_opaque.jsonl"))
m = re.search(r"in the file\. Opaque sources \(`r05final_configs_opaque\.jsonl`\): gamma-correct 4:1 [^\n]*\n[^\n]*\.", s)
s = s[:m.start()] + ("in the file. Opaque sources (`r05final_configs_opaque.jsonl`): gamma-correct 4:1 %.1f µs (%.1f %%), ×4 %.1f (%.1f %%), ratio 1.5 %.1f\n(%.1f %%), box %.1f (%.1f %%)."
                     % (op["SRGB"]["step_us"], 100 * op["SRGB"]["frac_of_8TBps"], op["SRGBUP"]["step_us"], 100 * op["SRGBUP"]["frac_of_8TBps"],
                        op["SRGBHALF"]["step_us"], 100 * op["SRGBHALF"]["frac_of_8TBps"], op["SRGBBOX"]["step_us"], 100 * op["SRGBBOX"]["frac_of_8TBps"])) + s[m.end():]
open(path, "w").write(s)
print("DESIGN.md section 4.1 rewritten from profiles/r05final_*")
