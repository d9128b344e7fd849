#!/usr/bin/env python3
"""Round 6: the headline (32 frames, one destination arena and four in rotation) for every ordered pair of kinds of memory -- source at the
head of a window of kind a, destination at the head of a window of kind b, windows by smol_hip_memory_zones() over one 240 GB region.
    python tools/exp_zone_matrix.py"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
gb = 240
W = 4 << 30
region = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
base = region.data_ptr()
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
n = 32
dst_bytes = n * slot_out


def headline(src_off, dst_off, rot):
    bs = []
    for k in range(rot):
        ims = [B._image(job, base + src_off + f * slot_in, base + dst_off + k * dst_bytes + f * slot_out, B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(n)]
        bs.append(sm.HipBatch(ims, device=0))
    r = sm.HipBatchRotation(bs)
    r.time(8, 8)
    t = min(r.time(0, 12) for _ in range(2)) / 12 * 1e3
    for b in bs:
        b.destroy()
    return t


headline(0, 8 << 30, 1)
z, kinds, c = sm.memory_zones(base, gb << 30, W)
print("windows of 4 GB:", "".join(str(v) for v in z), "kinds", kinds, "contrast %.3f" % c, flush=True)
# windows whose neighbour has the same label (their 4 GB are of one kind throughout, as far as the labels see)
solid = {k: [i for i in range(len(z) - 1) if z[i] == k and z[i + 1] == k] for k in range(kinds)}
for rot in (1, 4):
    print("destination arenas in rotation: %d   (rows: source kind, columns: destination kind; us per launch)" % rot)
    for a in range(kinds):
        row = []
        for b in range(kinds):
            cand = [(i, j) for i in solid[a] for j in solid[b] if i != j]
            if not cand:
                row.append("   --")
                continue
            ts = [headline(i * W, j * W, rot) for i, j in cand[:: max(1, len(cand) // 3)][:3]]
            row.append("%.1f" % min(ts) + "/%.1f" % max(ts))
        print("   source kind %d: " % a + "   ".join(row), flush=True)
