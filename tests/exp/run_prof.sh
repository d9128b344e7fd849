cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5mix -o c5mix -- python3 tools/bench_configs.py --configs C5mix --iters 10 > gpurun_out/prof_c5mix.log 2>&1
find gpurun_out/prof_c5mix -name "*kernel_stats*" | head
