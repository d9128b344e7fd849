timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/bench_host_path.py
python tools/bench_configs.py --iters 5 --warmup 1 --configs C5,C5mix --graph | cut -c1-14,100-400
