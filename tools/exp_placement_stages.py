#!/usr/bin/env python3
"""Round 6: does the picked destination keep its speed through the clean-up (candidates and spacers freed, cache emptied) and the
0.2 s clock settle that follow?  python tools/exp_placement_stages.py"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "free"
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
srcs = [torch.randint(0, 256, (32, slot_in), dtype=torch.uint8, device=dev) for _ in range(2)]
def mk(d): return [sm.HipBatch([B._image(job, src[f].data_ptr(), d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(32)], device=0) for src in srcs]
class Pair:
    def __init__(self, bs): self.bs = bs
    def time(self, w, n): return [b.time(w, n) for b in self.bs]
    def time_each(self, w, n): return self.bs[0].time_each(w, n)
def t(b, n=16): return "/".join("%.1f" % (x / n * 1e3) for x in b.time(1, n))
cands, spacers = [], []
for k in range(4):
    d = torch.zeros((32, slot_out), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
    b = Pair(mk(d))
    if k == 0: b.time(0, 500)
    cands.append((d, b))
    if k < 3: spacers.append(torch.empty(3 << 30, dtype=torch.uint8, device=dev))
print("probe 1      ", "  ".join(t(b) for d, b in cands), flush=True)
print("probe 2      ", "  ".join(t(b) for d, b in cands), flush=True)
print("probe 3 (x40)", "  ".join(t(b, 40) for d, b in cands), flush=True)
if mode == "free":
    del spacers
    print("spacers freed", "  ".join(t(b) for d, b in cands), flush=True)
    torch.cuda.empty_cache()
    print("cache emptied", "  ".join(t(b) for d, b in cands), flush=True)
for d, b in cands: b.time(0, 300)
print("after 4x300  ", "  ".join(t(b) for d, b in cands), flush=True)
print("x20 again    ", "  ".join("/".join("%.1f" % (x / 20 * 1e3) for x in b.time(0, 20)) for d, b in cands), flush=True)
print("x20 after 5  ", "  ".join("/".join("%.1f" % (x / 20 * 1e3) for x in b.time(5, 20)) for d, b in cands), flush=True)
time.sleep(0.5)
print("after sleep  ", "  ".join("/".join("%.1f" % (x / 20 * 1e3) for x in b.time(0, 20)) for d, b in cands), flush=True)
print("each         ", " | ".join(" ".join("%.0f" % (x * 1e3) for x in b.time_each(2, 8)) for d, b in cands), flush=True)
