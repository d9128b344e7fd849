run() { echo "== $*"; env "$@" python tools/bench_configs.py --iters 10 --warmup 2 --configs C3 2>&1 | grep -o '"step_us": [0-9.]*, .*"GBps": [0-9.]*'; }
run A=1
run SMOL_TILE_DBG=7
run SMOL_TILE_DBG=4
python tools/bench_configs.py --iters 10 --warmup 2 --configs C3,C4,C5mix --check | cut -c1-14,150-420
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
