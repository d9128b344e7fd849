cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04final > gpurun_out/r04final_round.log 2>&1
tail -5 gpurun_out/r04final_round.log
python tests/bench_sweep_with_reference.py --out gpurun_out/r04_sweep > gpurun_out/r04_sweep_log.txt 2>&1
tail -12 gpurun_out/r04_sweep_log.txt
