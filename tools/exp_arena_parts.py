#!/usr/bin/env python3
"""Round 6, after tools/exp_arena_candidates.py showed that the headline's mode belongs to the ARENA (its physical pages), not to the
process: (a) is it the source or the destination arena that decides? (source of arena i with the destination of arena j);
(b) is a slow arena slow everywhere? (its 32 frames in four groups of 8).     python tools/exp_arena_parts.py [K]"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
works = [B.work_c2(0, 1, dev, 0, 32) for _ in range(K)]
works[0].batch.time(0, 600)
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)


def batch_of(src, dst, frames):
    ims = [B._image(job, src[f].data_ptr(), dst[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in frames]
    return sm.HipBatch(ims, device=0)


def us(b, n):
    b.time(2, 10)
    return b.time(0, 40) / 40 * 1e3 / n * 32      # scaled to 32 frames


base = []
for k, w in enumerate(works):
    t = w.batch.time(2, 40) / 40 * 1e3
    base.append(t)
    print("arena %d: %.1f us per launch" % (k, t), flush=True)
print("(a) source arena i x destination arena j, us per launch")
for i in range(K):
    row = []
    for j in range(K):
        b = batch_of(works[i].keep[0], works[j].keep[1], range(32))
        row.append(us(b, 32))
        b.destroy()
    print("  src %d: %s" % (i, " ".join("%.1f" % x for x in row)), flush=True)
print("(b) frames 0-7 / 8-15 / 16-23 / 24-31 of each arena alone (8 frames per launch, scaled x4; 512 MB of source: partly cache-resident)")
for i in range(K):
    row = []
    for g in range(4):
        b = batch_of(works[i].keep[0], works[i].keep[1], range(8 * g, 8 * g + 8))
        row.append(us(b, 8))
        b.destroy()
    print("  arena %d (%.1f): %s" % (i, base[i], " ".join("%.1f" % x for x in row)), flush=True)
print("(c) frames 0-15 / 16-31 (16 frames per launch, scaled x2)")
for i in range(K):
    row = []
    for g in range(2):
        b = batch_of(works[i].keep[0], works[i].keep[1], range(16 * g, 16 * g + 16))
        row.append(us(b, 16))
        b.destroy()
    print("  arena %d (%.1f): %s" % (i, base[i], " ".join("%.1f" % x for x in row)), flush=True)
