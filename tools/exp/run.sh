run() { echo "== $*"; env "$@" python tools/bench_configs.py --iters 10 --warmup 2 --check --configs R1.5,R3,R6,PHOTO,PNG 2>&1 | grep -o '"config": "[A-Za-z0-9.]*"\|"step_us": [0-9.]*\|"GBps": [0-9.]*\|bit_exact[^}]*' | paste - - - - ; }
run A=1
run SMOL_TILE_CPL=4
run SMOL_TILE_CPL=4 SMOL_TILE_LDS_BUDGET=8192
