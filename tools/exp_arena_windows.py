#!/usr/bin/env python3
"""Round 6: the headline's mode is a property of the (source arena, destination arena) PAIR (tools/exp_arena_parts.py). Where do the
destinations that disagree with a source lie? One 4 GB destination region; the 128 MB destination arena placed at window offsets
of 64 MB across it, and at small shifts inside one window; K source arenas.     python tools/exp_arena_windows.py [K]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
import torch  # noqa: E402
import smolscale_amd as sm  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
works = [B.work_c2(0, 1, dev, 0, 32) for _ in range(K)]
works[0].batch.time(0, 600)
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
big = torch.zeros(4 << 30, dtype=torch.uint8, device=dev)
pitch_out = works[0].keep[1].stride(0)
print("destination region %#x, frame pitch %d" % (big.data_ptr(), pitch_out))


def us(src, dbase):
    ims = [B._image(job, src[f].data_ptr(), dbase + f * pitch_out, B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(32)]
    b = sm.HipBatch(ims, device=0)
    b.time(2, 6)
    t = b.time(0, 24) / 24 * 1e3
    b.destroy()
    return t


print("windows of 64 MB: one character per window, '#' = slow (> 332 us), '.' = fast")
for i in range(K):
    row = [us(works[i].keep[0], big.data_ptr() + (w << 26)) for w in range(62)]
    print("  src %d (own dst: %.1f): %s   min %.1f max %.1f" % (i, works[i].batch.time(2, 24) / 24 * 1e3,
                                                               "".join("#" if t > 332 else "." for t in row), min(row), max(row)), flush=True)
print("shifts inside window 0")
for i in range(K):
    row = []
    for sh in (0, 256, 4 << 10, 64 << 10, 1 << 20, 2 << 20, 4 << 20, 16 << 20, 32 << 20):
        row.append("%s=%.1f" % (sh, us(works[i].keep[0], big.data_ptr() + sh)))
    print("  src %d: %s" % (i, " ".join(row)), flush=True)
