#!/usr/bin/env python3
"""PCIe-inclusive timing of the drop-in host-pointer path (smol_scale_new +
smol_scale_batch on host buffers): upload of the needed rows, kernel, download.
Reported in DESIGN.md next to the device-resident number; never the benchmark value."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import smolscale_amd as sm  # noqa: E402


def run(devices, w_in, h_in, w_out, h_out, reps=10):
    if devices:
        os.environ["SMOL_HIP_DEVICES"] = devices
    else:
        os.environ.pop("SMOL_HIP_DEVICES", None)
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, w_in * h_in * 4, dtype=np.uint8)
    dst = np.zeros(w_out * h_out * 4, np.uint8)
    ctx = sm.SmolScaleCtx(src, 0, w_in, h_in, w_in * 4, dst, 0, w_out, h_out, w_out * 4, 0)
    ctx.batch(0, h_out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.batch(0, h_out)
        ts.append(time.perf_counter() - t0)
    ctx.destroy()
    t = sorted(ts)[len(ts) // 2]
    return {"devices": devices or "single", "job": "%dx%d->%dx%d RGBA8 premul, host pointers (pageable)" % (w_in, h_in, w_out, h_out),
            "median_ms": round(t * 1e3, 3), "Mpix_out_per_s": round(w_out * h_out / t / 1e6, 1),
            "host_GBps_in_plus_out": round((w_in * h_in * 4 + w_out * h_out * 4) / t / 1e9, 2)}


if __name__ == "__main__":
    for dev in ("", "0,0", "0,0,0,0"):
        print(json.dumps(run(dev, 4096, 4096, 1024, 1024)), flush=True)
    print(json.dumps(run("", 1920, 1080, 7680, 4320, reps=5)), flush=True)
