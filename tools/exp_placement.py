#!/usr/bin/env python3
"""Do the headline's two modes (profiles/r05_headline_modes.txt) follow the arena's ADDRESS, or only its physical placement?
One process; the 2 GB source arena + 128 MB destination arena are built repeatedly -- fresh from the driver (torch's cache
emptied), behind fillers of different sizes -- and the headline launch is timed for each build: virtual addresses and us per launch.
    python tools/exp_placement.py"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
fillers = [0, 0, 1 << 20, 2 << 20, 64 << 20, 256 << 20, 1 << 30, 3 << 20, 0, 17 << 20, 0, 513 << 20]
for k, fill in enumerate(fillers):
    torch.cuda.empty_cache()
    f = torch.empty(fill, dtype=torch.uint8, device=dev) if fill else None
    w = B.work_c2(0, 1, dev, 0, 32)
    src, dst = w.keep
    w.batch.time(0, 600)                      # settle the clock
    t = [w.batch.time(0, 50) / 50 * 1e3 for _ in range(3)]
    print("build %2d filler %4d MB  src %#x (mod 2 MB %#x)  dst %#x  us per launch %s" %
          (k, fill >> 20, src.data_ptr(), src.data_ptr() & ((2 << 20) - 1), dst.data_ptr(), " ".join("%.1f" % x for x in t)), flush=True)
    del w, src, dst, f
