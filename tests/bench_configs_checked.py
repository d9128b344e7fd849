#!/usr/bin/env python3
"""tools/bench_configs.py with a parity check: after timing, the first image of every config is
downloaded and its first rows are compared with the oracle (`bit_exact_first_rows_vs_oracle` in
the JSON line). Lives under tests/ because the oracle is test infrastructure.

    python tests/bench_configs_checked.py [--configs C1,C2,...] [--iters N] [--opaque] ...
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import helpers as H  # noqa: E402
import bench_configs  # noqa: E402


def oracle_checker(host_src, t_in, w_in, h_in, s_in, t_out, w_out, h_out, s_out, srgb, got):
    src = np.concatenate([host_src, np.zeros(16, np.uint8)])
    rows = min(h_out, 96)
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, s_in, s_out, first=0, n=rows)
    return np.array_equal(got[:rows * s_out], want)


if __name__ == "__main__":
    bench_configs.main([a for a in sys.argv[1:] if a != "--check"], checker=oracle_checker)
