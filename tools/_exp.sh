python tools/bench_configs.py --iters 5 --warmup 1 --configs P16UP,SRGBUP,P16HALF | cut -c1-18,130-330
SMOL_NO_WTILE=1 python tools/bench_configs.py --iters 5 --warmup 1 --configs P16UP,SRGBUP,P16HALF | cut -c1-18,130-330
