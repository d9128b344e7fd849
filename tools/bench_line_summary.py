"""one JSON bench line on stdin -> the headline's fraction, the placement candidates and the side table, one row each"""
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"): continue
    d = json.loads(ln)
    r = d["roofline"]
    print("frac %.4f launch %.1f us placement %s ok %s" % (r["frac"], r["launch_us"], (d.get("placement") or {}).get("candidates_us"), d["verified_vs_oracle"]["all_ranks_ok"]))
    for k, v in (d.get("other_baseline_configs") or {}).items():
        print("   %-8s %7.1f us %.3f  %s" % (k.split()[0], v.get("step_us", 0), v.get("frac_of_peak", 0), (v.get("placement") or {}).get("candidates_us")))
