timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"value": [0-9.]*\|"achieved": [0-9.]*\|"frac": [0-9.]*\|"launch_us": [0-9.]*' | paste - - - -
python tests/bench_configs_checked.py --iters 10 --warmup 2 --configs C1,C2,C3,C4,C5,BOX,R1.5,R3,R6,PHOTO,PNG,C5mix,P16,SRGB 2>&1 | grep -o '"config": "[A-Za-z0-9.]*"\|"step_us": [0-9.]*\|"GBps": [0-9.]*\|bit_exact[^}]*' | paste - - - -
