#!/usr/bin/env python3
"""C5mix's two timing modes (profiles/r04_c5mix_modes.txt): is the mode a property of the image ARENAS or of what the batch
allocates for itself (its table of job descriptors)? One process: the arenas are allocated once; the batch is created,
timed and destroyed several times over the same arenas (a held allocation of another size in between moves the table);
then the arenas are re-allocated and the same is done again.

    python tools/c5mix_batch_vs_arena.py [arena_rounds=3] [batches_per_arena=4]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import smolscale_amd as sm  # noqa: E402

arena_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
per_arena = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, W_IN, W_OUT = 10000, 256, 64
slot_in = ((W_IN * W_IN * 4 + 64 + 255) // 256) * 256
slot_out = ((W_OUT * W_OUT * 4 + 64 + 255) // 256) * 256
trng = np.random.default_rng(5)
types = [(int(trng.integers(0, 10)), int(trng.integers(0, 10))) for _ in range(N)]
host = np.random.default_rng(1).integers(0, 256, W_IN * W_IN * 4, dtype=np.uint8)
keep = []
for ar in range(arena_rounds):
    arena_in, arena_out = sm.DeviceBuffer(N * slot_in), sm.DeviceBuffer(N * slot_out)
    for i in range(0, N, 625):                      # (content does not matter for the timing; a few slots get real bytes)
        arena_in.upload(host, offset=i * slot_in)
    images = []
    for i, (t_in, t_out) in enumerate(types):
        im = sm.SmolHipImage()
        im.job = sm.plan_job(t_in, W_IN, W_IN, t_out, W_OUT, W_OUT, 0)
        im.src, im.dst = arena_in.ptr + i * slot_in, arena_out.ptr + i * slot_out
        im.rowstride_in, im.rowstride_out = W_IN * sm.bpp(t_in) if hasattr(sm, "bpp") else W_IN * (3 if t_in >= 8 else 4), W_OUT * (3 if t_out >= 8 else 4)
        im.first_row, im.n_rows = 0, W_OUT
        images.append(im)
    for b in range(per_arena):
        batch = sm.HipBatch(images)
        batch.time(200, 5)
        us = batch.time(5, 20) / 20 * 1e3
        print(json.dumps({"arena_round": ar, "arena_in": "%x" % arena_in.ptr, "batch": b, "step_us": round(us, 1)}), flush=True)
        batch.destroy()
        keep.append(sm.DeviceBuffer((1 + (7 * b + 3 * ar) % 11) << 20))
        if len(keep) > 2:
            keep.pop(0).free()
    arena_in.free()
    arena_out.free()
    keep.append(sm.DeviceBuffer((5 + ar) << 20))
