cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
python tools/r04_sweep_dbg.py 2>&1 | tail -2
python -m pytest tests/test_hip_parity.py tests/test_hip_multi.py tests/test_verify_suites.py -m gpu -x -q 2>&1 | tail -3
python tools/fuzz/gpu_fuzz.py 51 8000 > gpurun_out/r04f/fuzz.txt 2>&1; tail -3 gpurun_out/r04f/fuzz.txt | cut -c1-600
