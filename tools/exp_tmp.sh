cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 tools/fuzz/gpu_fuzz.py 813 40000 2>&1 | tail -2 | cut -c1-300
