"""Markdown table of a round's config rows beside an earlier round's:
    python tools/configs_table.py profiles/r04final_configs.jsonl [profiles/r03final_configs.jsonl]"""
import json, sys

def rows(path):
    out = {}
    for line in open(path):
        try:
            d = json.loads(line)
        except ValueError:
            continue
        out[d["config"]] = d
    return out

new = rows(sys.argv[1])
old = rows(sys.argv[2]) if len(sys.argv) > 2 else {}
print("| config | what | kernel | step | % of 8 TB/s | before |")
print("|---|---|---|---|---|---|")
for name, d in new.items():
    o = old.get(name)
    before = "%.1f µs, %.1f %%" % (o["step_us"], 100.0 * o["frac_of_8TBps"]) if o else "—"
    print("| %s | %s | `%s` | %.1f µs | %.1f %% | %s |" % (name, d.get("desc", d.get("description", "")), d.get("kernel", ""),
                                                        d["step_us"], 100.0 * d["frac_of_8TBps"], before))
