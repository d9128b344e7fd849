cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/gpu_tests_final.txt
cat gpurun_out/gpu_tests_final.txt
python3 tools/fuzz/gpu_fuzz.py 77 30000 2>&1 | tail -3 > gpurun_out/fuzz_after_up24.txt
cat gpurun_out/fuzz_after_up24.txt
