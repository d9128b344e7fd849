#!/bin/bash
# End-of-round evidence run (GPU box). Everything the bench line's fractions rest on is re-derivable from what this
# script leaves under profiles/ (tools/profile_summary.py condenses, tools/roofline_check.py re-derives and compares):
#   - the bench line itself (N=1, default flags)
#   - rocprofv3 --kernel-trace of the HEADLINE ALONE (--no-other-configs: one workload per kernel symbol) and one
#     pass per side config of the line (tools/bench_configs.py --configs X), so that no average mixes two workloads
#   - HBM traffic: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md), headline and side configs
#   - SQ counters (LDS bank conflicts, instruction mix) of the BOX row
# Usage: bash tools/profile_round.sh <tag>        (no --pmc together with trace domains other than --kernel-trace)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
LINE_CFGS="C1 C3 C4 C5 C5mix BOX C2CROP4"
TABLE=C1,C2,C3,C4,C5,C5mix,BOX,C2CROP4,C2ODD,R1.03,R1.1,R1.2,R1.35,R1.5,R1.7,R1.9,R2.5,R3,R3.5,R5,R6,R7,U1.1,U1.37,U1.5,U2,U2.6,U3,PHOTO,PHOTOODD,PNG,R1.5RGB,R1.5U,HDOWNVUP,THUMB,P16,SRGB,P16UP,P16HALF,SRGBUP,SRGBPNGUP,SRGBHALF,SRGBPHOTO,SRGBPNG,SRGBTHUMB,SRGBBOX,BOX128,WIDEBOX,SRGBHDVU,SRGBVCOPY,SRGBBOXBIL,P16BOXBIL,P16HDVU,S4K109ODD,S4K91ODD,SRGBUPODD,P16UPODD,P16HALFODD,SRGBBOX64,SRGBBOX128,FLATBOXBIL,FLATBILBOX,FLATTALL,SRGBU1.1,SRGBU2,SRGBUP24,BOX12,BOX16,U8,U4RGB,U2UNASSOC,SRGBR6,SRGBR7RGB,SRGBBILBOX,P16BILBOX,P16LBILBOX
python3 bench.py --steps 50 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
# the two measurements BASELINE itself names (Next 3): config 5 as a hipGraph of per-image kernel nodes beside the
# descriptor-table launch (graph_mode), config 4 with its PCIe-inclusive leg (e2e: page-locked and pageable)
python3 bench.py --config c5mix --steps 20 --warmup 5 > $O/c5mix_line.json 2> $O/c5mix_err.txt
python3 bench.py --config c4 --steps 20 --warmup 5 > $O/c4_line.json 2> $O/c4_err.txt
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o b --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-e2e > $O/kt_bench.log 2>&1
for c in $LINE_CFGS; do
  # (--placement far: sources and destinations as windows of one region, 32 GB apart -- what bench.py's placement probe finds:
  # profiles/r06_headline_modes.txt)
  rocprofv3 --kernel-trace --stats -d $O/kt_cfg_$c -o c --output-format csv -- python3 tools/bench_configs.py --iters 20 --warmup 3 --placement far --configs $c > $O/kt_cfg_$c.log 2>&1
done
SMOL_BENCH_GPU_STATE=1 python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --placement far --configs $TABLE > $O/configs.jsonl 2> $O/configs_err.txt
python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --placement far --opaque --configs SRGB,SRGBUP,SRGBHALF,SRGBPNG,SRGBBOX,C5mix > $O/configs_opaque.jsonl 2>> $O/configs_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_bench_$c -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-e2e > $O/pmc_bench_$c.log 2>&1
  for k in $LINE_CFGS R3; do
    rocprofv3 --pmc $c -d $O/pmc_cfg_${k}_$c -o p --output-format csv -- python3 tools/bench_configs.py --iters 5 --warmup 1 --settle 0.05 --placement far --configs $k > $O/pmc_cfg_${k}_$c.log 2>&1
  done
done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/sq_box/p$i -o p --output-format csv -- python3 tools/bench_configs.py --configs BOX --iters 3 --warmup 1 --settle 0.05 > $O/sq_box_log$i.txt 2>&1 || echo "SQ set $i failed"
done
# SQ counters of the gamma-correct rows (128bpp class: vector issue against table look-ups)
for c in SRGB SRGBUP SRGBHALF SRGBPHOTO SRGBBOX; do bash tools/sq_counters.sh $TAG $c > /dev/null 2>&1; done
# (gpurun merges gpurun_out/ back, not profiles/: run the two lines below again in the build container afterwards)
python3 tools/profile_summary.py $TAG
python3 tools/roofline_check.py $TAG | tee $O/roofline_check.txt
cp $O/roofline_check.txt profiles/${TAG}_roofline_check.txt
cp $O/c5mix_line.json profiles/${TAG}_c5mix_line.json; cp $O/c4_line.json profiles/${TAG}_c4_line.json
