#!/bin/bash
# SQ instruction / wait counters for one config of tools/bench_configs.py (separate --pmc passes):
#   bash tools/sq_counters.sh <config> <tag>   ->  gpurun_out/<tag>/sq_<config>.txt
CFG=$1; TAG=${2:-sq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/sq_$CFG/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs $CFG --iters 3 --warmup 1 > $O/sq_${CFG}_log$i.txt 2>&1 || echo "SQ set $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/sq_$CFG/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/sq_$CFG.txt","w") as fh:
    for k,v in acc.items():
        if "smolhip" not in k: continue
        fh.write(k+"\n")
        for c,x in sorted(v.items()): fh.write("   %-26s n=%d per launch %.6g\n" % (c,len(x),sum(x)/len(x)))
print(open("$O/sq_$CFG.txt").read())
PY
