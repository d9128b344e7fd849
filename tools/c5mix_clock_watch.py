#!/usr/bin/env python3
"""C5mix timed for ~15 s in one process, one line per 200 launches, next to the clocks rocm-smi reports at that moment:
does the launch time follow a clock (sclk / fclk / mclk power state) rather than anything in the kernel?
    python tools/c5mix_clock_watch.py [config=C5mix] [lines=40]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import smolscale_amd as sm
import bench_configs as B

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5mix"
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 40
# build the batch once (the body of run_config, without its timing)
t_in0, w_in, h_in, stride_in, t_out0, w_out, h_out, n_images, desc = B.CONFIGS[cfg][:9]
trng = np.random.default_rng(20260202)
slot_in = ((h_in * w_in * 4 + 64 + 255) // 256) * 256
slot_out = ((h_out * w_out * 4 + 64 + 255) // 256) * 256
a_in, a_out = sm.DeviceBuffer(n_images * slot_in), sm.DeviceBuffer(n_images * slot_out)
a_in.fill(0x5a)
images = []
for i in range(n_images):
    t_in = int(trng.integers(0, 10)) if t_in0 < 0 else t_in0
    t_out = int(trng.integers(0, 10)) if t_out0 < 0 else t_out0
    im = sm.SmolHipImage()
    im.job = sm.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
    im.src, im.dst = a_in.ptr + i * slot_in, a_out.ptr + i * slot_out
    im.rowstride_in, im.rowstride_out = w_in * B.bpp(t_in), w_out * B.bpp(t_out)
    im.first_row, im.n_rows = 0, h_out
    images.append(im)
batch = sm.HipBatch(images)
def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        keep = [l.split(":", 2)[-1].strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power (W)", "junction", "hbm"))]
        return " | ".join(keep)
    except Exception as e:
        return repr(e)
for i in range(lines):
    ms = batch.time(0, 200) / 200
    print("%2d  %.1f us   %s" % (i, ms * 1e3, clocks()), flush=True)
