#!/bin/bash
# End-of-round evidence run (GPU box): rocprofv3 kernel-trace stats for the bench line and for every config,
# HBM traffic counters (separate --pmc passes, MI355X_MICROARCH.md) for the headline kernel and the other
# BASELINE configs, SQ instruction counters for config 4. Usage: bash tools/profile_round.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
CFGS=C1,C2,C3,C4,C5,C5mix,BOX,R1.5,R3,R6,PHOTO,PNG,P16,SRGB,P16UP,P16HALF,SRGBUP,SRGBPNGUP,SRGBHALF,SRGBPHOTO,SRGBPNG,THUMB,SRGBTHUMB,SRGBBOX,BOX128
python3 bench.py --steps 50 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o b --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/kt_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_configs -o c --output-format csv -- python3 tools/bench_configs.py --iters 10 --warmup 2 --configs $CFGS > $O/kt_configs.log 2>&1
python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --configs $CFGS > $O/configs.jsonl 2> $O/configs_err.txt
python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --opaque --configs SRGB,SRGBUP,SRGBHALF,SRGBPNG,SRGBBOX,C5mix > $O/configs_opaque.jsonl 2>> $O/configs_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_bench_$c -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $O/pmc_bench_$c.log 2>&1
  rocprofv3 --pmc $c -d $O/pmc_cfg_$c -o p --output-format csv -- python3 tools/bench_configs.py --iters 5 --warmup 1 --configs C3,C4,C5mix,BOX,R3 > $O/pmc_cfg_$c.log 2>&1
done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/sq_c4/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs C4 --iters 3 --warmup 1 > $O/sq_c4_log$i.txt 2>&1 || echo "SQ set $i failed"
done
python3 tools/profile_summary.py $TAG
