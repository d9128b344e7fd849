cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_thumb -o t --output-format csv -- python3 tools/bench_configs.py --configs THUMB,SRGBTHUMB --iters 10 --settle 0.05 > gpurun_out/prof_thumb.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_thumb/t_kernel_stats.csv")):
    if "smolhip" in r["Name"]: print(r["Name"][14:60], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
