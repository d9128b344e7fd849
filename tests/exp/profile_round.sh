# End-of-round evidence run (GPU box): kernel-trace stats for the bench line and for every config,
# HBM traffic counters (separate --pmc passes) for the headline kernel and for k_up / k_pow2x.
TAG=${1:-r01e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python3 bench.py --steps 50 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o b --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/kt_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_configs -o c --output-format csv -- python3 tools/bench_configs.py --iters 10 --warmup 2 --configs C1,C2,C3,C4,C5,C5mix,BOX,R1.5,R3,R6,PHOTO,PNG,P16,SRGB,P16UP,P16HALF,SRGBUP,SRGBHALF,SRGBPHOTO,THUMB,SRGBTHUMB > $O/kt_configs.log 2>&1
python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --configs C1,C2,C3,C4,C5,C5mix,BOX,R1.5,R3,R6,PHOTO,PNG,P16,SRGB,P16UP,P16HALF,SRGBUP,SRGBPNGUP,SRGBHALF,SRGBPHOTO,SRGBPNG,THUMB,SRGBTHUMB,SRGBBOX,BOX128 > $O/configs.jsonl 2> $O/configs_err.txt
python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --opaque --configs SRGB,SRGBUP,SRGBHALF,SRGBPNG,SRGBBOX,C5mix > $O/configs_opaque.jsonl 2>> $O/configs_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_bench_$c -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_bench_$c.log 2>&1
  rocprofv3 --pmc $c -d $O/pmc_cfg_$c -o p --output-format csv -- python3 tools/bench_configs.py --iters 5 --warmup 1 --configs C3,C4,C5mix,P16 > $O/pmc_cfg_$c.log 2>&1
done
find $O -name "*.csv" | head -30
