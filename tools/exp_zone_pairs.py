#!/usr/bin/env python3
"""Round 6: are the odd windows of smol_hip_memory_zones()'s first sweep a matter of the two streams' RELATIVE position? Window 0 beside
every 2 GB window of the first 80 GB of a 240 GB region, 256 MB of each, with the reference stream started at three offsets inside window 0.
    python tools/exp_zone_pairs.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
zp = ctypes.CDLL(os.path.join(ROOT, "tools/microbench/libzone_probe.so"))
zp.smol_zone_probe_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
region = torch.empty(240 << 30, dtype=torch.uint8, device=dev)
base = region.data_ptr()
W = 2 << 30
P = 256 << 20


def pair(a, b, iters=8):
    us = ctypes.c_double()
    assert zp.smol_zone_probe_pair(base + a, base + b, P, 2, 1, iters, ctypes.byref(us)) == 0
    return us.value


for _ in range(200):
    pair(0, P)
z, kinds, c = sm.memory_zones(base, 240 << 30, W)
print("library:", "".join(str(v) for v in z[:40]), kinds, "%.3f" % c)
offs = (0, (512 << 20) + 4352, (1 << 30) + (256 << 20) + 8704, (1792 << 20) - 4096)
same = min(pair(0, P) for _ in range(3))
print("halves of window 0: %.1f us" % same)
for o in offs:
    row = [min(pair(o, w * W), pair(o, w * W)) / same for w in range(1, 40)]
    print("ref at +%4d MB: " % (o >> 20) + " ".join("%.2f" % v for v in row))
for o in (P // 2 + 4352, P - 8704):
    row = [min(pair(0, w * W + o), pair(0, w * W + o)) / same for w in range(1, 40)]
    print("window at +%4d MB: " % (o >> 20) + " ".join("%.2f" % v for v in row))
