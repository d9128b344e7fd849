timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5mix -- python3 $R/tools/bench_configs.py --iters 10 --warmup 2 --configs C5mix > $R/gpurun_out/prof_c5mix.log 2>&1
cd $R; cut -c1-140 gpurun_out/prof_c5mix/*/*kernel_stats.csv | head -12
python tools/bench_configs.py --iters 10 --warmup 2 --check | cut -c1-16,150-420
