#!/usr/bin/env python3
"""What the drop-in costs a caller that asks for its rows a few at a time (the reference's own outer loop is one output
row per step, smolscale.c:507-546): ONE context per image, smol_scale_batch() for n rows at a time until the image is
done, host memory, PCIe inclusive. One JSON line per (geometry, memory, rows per call); `x_whole` = time relative to
one call for the whole image. Also importable: bench.py puts the same table into its `e2e` object.

    python tools/bench_rows_per_call.py [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import smolscale_amd as sm  # noqa: E402

GEOMETRIES = {
    "1080p_to_720p": (0, 1920, 1080, 0, 1280, 720),
    "c2_4096_to_1024": (0, 4096, 4096, 0, 1024, 1024),
}


def measure(geom, pinned, rows_per_call, reps=3, want=None):
    t_in, w_in, h_in, t_out, w_out, h_out = geom
    bi, bo = sm.bytes_per_pixel(t_in), sm.bytes_per_pixel(t_out)
    n_in, n_out = w_in * h_in * bi, w_out * h_out * bo
    rng = np.random.default_rng(7)
    keep = []
    if pinned:
        a, b = sm.PinnedArray(n_in), sm.PinnedArray(n_out)
        keep = [a, b]
        src, dst = a.array, b.array
        src[:] = rng.integers(0, 256, n_in, dtype=np.uint8)
    else:
        src = rng.integers(0, 256, n_in, dtype=np.uint8)
        dst = np.zeros(n_out, np.uint8)
    step = h_out if rows_per_call in (0, None) else rows_per_call
    lib = sm.lib()
    best = 1e9
    with sm.SmolScaleCtx(src, t_in, w_in, h_in, w_in * bi, dst, t_out, w_out, h_out, w_out * bo, 0) as ctx:
        c = ctx._ctx
        ctx.batch(0, h_out)                           # warm-up: lanes, rings, slabs
        whole = dst.copy()
        for _ in range(reps):
            dst[:] = 0
            t0 = time.perf_counter()
            for y in range(0, h_out, step):
                lib.smol_scale_batch(c, y, min(step, h_out - y))
            best = min(best, time.perf_counter() - t0)
        same = bool(np.array_equal(dst, whole))
    for k in keep:
        k.free()
    return {"ms_per_image": round(best * 1e3, 3), "us_per_call": round(best * 1e6 / ((h_out + step - 1) // step), 1),
            "equal_to_whole_image_call": same}


def table(reps=3, rows=(1, 8, 64, 0)):
    out = {}
    for name, geom in GEOMETRIES.items():
        for mem in ("pageable", "pinned"):
            per = {}
            for r in rows:
                per["all" if r == 0 else str(r)] = measure(geom, mem == "pinned", r, reps)
            whole = per["all"]["ms_per_image"]
            for v in per.values():
                v["x_whole"] = round(v["ms_per_image"] / whole, 2)
            out["%s.%s" % (name, mem)] = per
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    for k, v in table(args.reps).items():
        print(json.dumps({"rows_per_call": k, **v}), flush=True)
