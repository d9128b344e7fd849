cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_env.sh SRGBUPODD,P16UPODD,P16HALFODD,SRGBUP,P16UP,P16HALF -- "SMOL_TILE_WT=1" "X=1"
python tools/bench_sweep.py --out gpurun_out/r04_sweep3 > gpurun_out/r04_sweep3_log.txt 2>&1
tail -4 gpurun_out/r04_sweep3_log.txt
