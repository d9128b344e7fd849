"""one JSON bench line on stdin -> the headline's fraction, the placement probes and the side table, one row each"""
import sys, json


def pl(p):
    if not p or not p.get("src_windows_us"):
        return "-"
    return "src %s @%s GB  dst %s" % (" ".join("%.0f" % x for x in p["src_windows_us"]), p.get("src_offset_GB"), " ".join("%.0f" % x for x in p["dst_windows_us"]))


for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r = d["roofline"]
    sd = r.get("same_destination") or {}
    print("frac %.4f launch %.1f us (x%s arenas; same destination: %s us, %s) ok %s  %s" % (r["frac"], r["launch_us"], r.get("destination_arenas"), sd.get("launch_us"), sd.get("frac"),
                                                                                         d["verified_vs_oracle"]["all_ranks_ok"], pl(d.get("placement"))))
    for k, v in (d.get("other_baseline_configs") or {}).items():
        print("   %-8s %7.1f us %.3f  x%s  %s" % (k.split()[0], v.get("step_us", 0), v.get("frac_of_peak", 0), (v.get("placement") or {}).get("dst_rotation"), pl(v.get("placement")) if "error" not in v else v["error"]))
