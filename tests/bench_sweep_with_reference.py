#!/usr/bin/env python3
"""tools/bench_sweep.py with the reference's own rows beside the library's: `smol' (one
thread) and `smol-mt' (every host thread, one output-row strip per thread -- test.c:838-883) from the REAL
reference built by oracle/Makefile (oracle/_ref, AVX2 build), timed on the same box on the same frame and
the same 200 output sizes, in the reference's samples / average text formats (bench.sh:39-71,
test.c:343-382). Lives under tests/ because it loads oracle/_ref; tools/ stays oracle-free.

    python tests/bench_sweep_with_reference.py [--out DIR] [--tests ...] [--steps N]
"""
import ctypes
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
import bench_sweep  # noqa: E402

ARGB_P = 2


def reference_rows(src, w_in, h_in, sizes, reps):
    if not H.have_ref("avx2"):
        return {}
    lib = H.ref("avx2")
    lib.ref_probe_mt_time.restype = ctypes.c_double
    lib.ref_probe_mt_time.argtypes = [ctypes.c_void_p] + [ctypes.c_uint32] * 4 + [ctypes.c_void_p] + [ctypes.c_uint32] * 4 + [ctypes.c_uint32] * 3
    n_threads = os.cpu_count() or 1
    single, multi = [], []
    reps = min(reps, 2)                 # the single-threaded 8K rows alone would take ten minutes at 5 repetitions
    srcp = np.concatenate([src, np.zeros(16, np.uint8)])
    for (w_out, h_out) in sizes:
        dst = np.empty(w_out * h_out * 4 + 16, np.uint8)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            lib.smol_scale_simple(srcp.ctypes.data, ARGB_P, w_in, h_in, w_in * 4, dst.ctypes.data, ARGB_P, w_out, h_out, w_out * 4, 0)
            best = min(best, time.perf_counter() - t0)
        single.append((w_out, h_out, best, (w_in * h_in + w_out * h_out) / best))
        best = 1e9
        for _ in range(reps):
            best = min(best, lib.ref_probe_mt_time(srcp.ctypes.data, ARGB_P, w_in, h_in, w_in * 4, dst.ctypes.data, ARGB_P,
                                                   w_out, h_out, w_out * 4, 0, min(n_threads, h_out), 1))
        multi.append((w_out, h_out, best, (w_in * h_in + w_out * h_out) / best))
    return {"smol (reference, AVX2, 1 thread)": single, "smol-mt (reference, AVX2, %d threads)" % n_threads: multi}


if __name__ == "__main__":
    bench_sweep.main(reference_rows=reference_rows)
