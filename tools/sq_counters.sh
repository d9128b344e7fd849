#!/bin/bash
# SQ instruction / LDS counters of one config of tools/bench_configs.py (GPU box):
#   bash tools/sq_counters.sh <tag> <config> [extra bench_configs.py flags, e.g. --opaque]
# leaves gpurun_out/<tag>/sq_<config>.txt (mean per launch per kernel; tools/sq_summary.py)
TAG=$1; CFG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/sq_$CFG/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs $CFG --iters 3 --warmup 1 --settle 0.05 "$@" > $O/sq_${CFG}_log$i.txt 2>&1 || echo "SQ set $i failed ($set)"
done
python3 tools/sq_summary.py $O/sq_$CFG > $O/sq_$CFG.txt
cat $O/sq_$CFG.txt
