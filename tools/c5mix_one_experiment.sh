#!/bin/bash
# VERDICT round 4, Next 8: ONE experiment on C5mix's two timing modes. N fresh processes; each runs C5mix under
# rocprofv3 with the kernel trace (durations) AND the L2 counters of the same dispatches, so that a fast and a slow
# process can be compared counter by counter. Output: one line per process (mean duration of the mixed-type kernel over
# its last dispatches, TCC hits / misses / HBM reads per launch), gpurun_out/<tag>/c5mix_modes.txt
TAG=${1:-r05_c5mix}; N=${2:-12}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $N); do
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $O/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs C5mix --iters 10 --warmup 2 --settle 0.1 > $O/log$i.txt 2>&1 || echo "run $i failed"
done
python3 - "$O" "$N" <<'PY'
import csv, sys, glob, os, json
O, N = sys.argv[1], int(sys.argv[2])
out = []
for i in range(1, N + 1):
    d = os.path.join(O, "p%d" % i)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    line = {"process": i}
    try:
        step = json.loads(open(os.path.join(O, "log%d.txt" % i)).read().strip().splitlines()[-1])
        line["step_us_events"] = step["step_us"]
    except Exception as e:
        line["step_us_events"] = None
    if kt:
        rows = [r for r in csv.DictReader(open(kt[0])) if "pow2x" in r["Kernel_Name"]]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-8:]
        line["kernel_us_last8"] = round(sum(dur) / max(1, len(dur)), 1)
    if cc:
        acc = {}
        for r in csv.DictReader(open(cc[0])):
            if "pow2x" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            line[k] = round(sum(v) / len(v))
    out.append(line)
with open(os.path.join(O, "c5mix_modes.txt"), "w") as f:
    for l in out:
        f.write(json.dumps(l) + "\n")
        print(json.dumps(l))
PY
