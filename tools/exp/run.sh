timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python tools/bench_configs.py --iters 5 --warmup 1 --check --configs BOX128 | cut -c1-18,100-135,150-330
