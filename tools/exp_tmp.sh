cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python3 -m pytest tests/test_hip_multi.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
echo "## SMOL_HIP_RESIDENT_SRC=1"; SMOL_HIP_RESIDENT_SRC=1 python3 tools/bench_rows_per_call.py --reps 3
echo "## default"; python3 tools/bench_rows_per_call.py --reps 3
done
