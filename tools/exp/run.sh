timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python tools/bench_configs.py --iters 10 --warmup 2 --check --configs C3,C4,BOX,R3,PHOTO,PNG,C5mix | cut -c1-18,150-330
