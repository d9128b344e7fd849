#!/usr/bin/env python3
"""Mean per launch of every counter rocprofv3 --pmc left under a directory, per kernel (tools/sq_counters.sh)."""
import collections, csv, glob, os, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in out.items():
    if "smolhip" not in k:
        continue
    print(k[:120])
    m = {c: sum(x) / len(x) for c, x in v.items()}
    for c in sorted(m):
        print("   %-28s n=%d  per launch %.6g" % (c, len(v[c]), m[c]))
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        print("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
    if "GRBM_GUI_ACTIVE" in m:
        g = m["GRBM_GUI_ACTIVE"]
        if "SQ_LDS_IDX_ACTIVE" in m:
            print("   LDS active cycles per CU / kernel cycles = %.3f" % (m["SQ_LDS_IDX_ACTIVE"] / 256.0 / g))
        if "SQ_ACTIVE_INST_VALU" in m:
            print("   VALU active cycles per SIMD / kernel cycles = %.3f  (SQ_ACTIVE_INST_VALU / 1024 / GRBM_GUI_ACTIVE, x4 if the counter is in quad-cycles)" % (m["SQ_ACTIVE_INST_VALU"] / 1024.0 / g))
