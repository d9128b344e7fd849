timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 100 --warmup 10 --no-cpu-baseline | cut -c1-120,560-900
python tools/bench_configs.py --iters 10 --warmup 2 --check | cut -c1-14,150-420
