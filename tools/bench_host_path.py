#!/usr/bin/env python3
"""PCIe-inclusive timing of the drop-in host-pointer path (smol_scale_new + smol_scale_batch on host buffers): upload
of the needed rows, kernels, download. Reported in DESIGN.md next to the device-resident number; never the benchmark
value.

    python tools/bench_host_path.py                          the round-2 table (device lists 0 / 0,0 / 0,0,0,0)
    python tools/bench_host_path.py --devices all --memory pinned|pageable
                                                             one JSON line: C2 and C4 geometry with SMOL_HIP_DEVICES
                                                             set as given (tools/scale_curve.sh)
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


def run(devices, t_in, w_in, h_in, t_out, w_out, h_out, reps=10, pinned=False):
    if devices:
        os.environ["SMOL_HIP_DEVICES"] = devices
    else:
        os.environ.pop("SMOL_HIP_DEVICES", None)
    bi, bo = sm.bytes_per_pixel(t_in), sm.bytes_per_pixel(t_out)
    rng = np.random.default_rng(1)
    n_in, n_out = w_in * h_in * bi, w_out * h_out * bo
    keep = []
    if pinned:
        a, b = sm.PinnedArray(n_in), sm.PinnedArray(n_out)
        keep = [a, b]
        src, dst = a.array, b.array
        src[:] = rng.integers(0, 256, n_in, dtype=np.uint8)
    else:
        src = rng.integers(0, 256, n_in, dtype=np.uint8)
        dst = np.zeros(n_out, np.uint8)
    ctx = sm.SmolScaleCtx(src, t_in, w_in, h_in, w_in * bi, dst, t_out, w_out, h_out, w_out * bo, 0)
    ctx.batch(0, h_out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.batch(0, h_out)
        ts.append(time.perf_counter() - t0)
    ctx.destroy()
    for k in keep:
        k.free()
    t = sorted(ts)[len(ts) // 2]
    return {"devices": devices or "single", "memory": "pinned" if pinned else "pageable",
            "job": "%dx%d (type %d) -> %dx%d (type %d), host pointers" % (w_in, h_in, t_in, w_out, h_out, t_out),
            "median_ms": round(t * 1e3, 3), "Mpix_out_per_s": round(w_out * h_out / t / 1e6, 1),
            "host_GBps_in_plus_out": round((n_in + n_out) / t / 1e9, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None)
    ap.add_argument("--memory", default="pageable", choices=["pinned", "pageable"])
    args = ap.parse_args()
    if args.devices is not None:
        pin = args.memory == "pinned"
        print(json.dumps({"host_path": [run(args.devices, 0, 4096, 4096, 0, 1024, 1024, pinned=pin),
                                        run(args.devices, 8, 65535, 4096, 0, 8192, 4096, reps=3, pinned=pin)]}), flush=True)
    else:
        for dev in ("", "0,0", "0,0,0,0"):
            print(json.dumps(run(dev, 0, 4096, 4096, 0, 1024, 1024)), flush=True)
        print(json.dumps(run("", 0, 1920, 1080, 0, 7680, 4320, reps=5)), flush=True)
