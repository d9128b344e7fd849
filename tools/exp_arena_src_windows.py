#!/usr/bin/env python3
"""Round 6: the headline's source arena as WINDOWS of one large region (default 48 GB) -- does the mode follow the window's position inside
one allocation, as it follows the allocation? One destination arena (and a second one for comparison).
    python tools/exp_arena_src_windows.py [region GB] [step MB]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
step_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048 + 64
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
dsts = [torch.zeros((32, slot_out), dtype=torch.uint8, device=dev) for _ in range(2)]
big = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
for off in range(0, gb << 30, 1 << 30):          # fill in 1 GB pieces (random bytes)
    big[off:off + (1 << 30)].random_(0, 256)
torch.cuda.synchronize()
print("region %#x (%d GB), destinations %#x %#x" % (big.data_ptr(), gb, dsts[0].data_ptr(), dsts[1].data_ptr()), flush=True)


def us(base, d):
    ims = [B._image(job, base + f * slot_in, d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(32)]
    b = sm.HipBatch(ims, device=0)
    b.time(2, 6)
    t = b.time(0, 16) / 16 * 1e3
    b.destroy()
    return t


b0 = sm.HipBatch([B._image(job, big.data_ptr() + f * slot_in, dsts[0][f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(32)], device=0)
b0.time(0, 500)
need = 32 * slot_in
offs = list(range(0, (gb << 30) - need, step_mb << 20))
for d in range(2):
    row = [us(big.data_ptr() + o, dsts[d]) for o in offs]
    print("dst %d: %s" % (d, " ".join("%.0f" % t for t in row)), flush=True)
    print("        %s   (# = slow > 331 us)" % "".join("#" if t > 331 else "." for t in row), flush=True)
