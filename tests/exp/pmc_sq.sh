# SQ counters for one config (separate passes, --pmc only; see MI355X_MICROARCH.md)
CFG=${1:-C4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d gpurun_out/pmc_$CFG/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs $CFG --iters 3 --warmup 1 > gpurun_out/pmc_$CFG/log$i.txt 2>&1 || echo "set $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_$CFG/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-28s %.4g" % (c,x))
PY
