"""Randomized parity run of the road from HOST memory (strip_issue_locked, smol_rt_lanes.hip) against the oracle: random jobs of every
class (tests' random_case), page-locked (allocated or registered in place) or pageable memory on either side, padded rows, a
byte-misaligned source, a row range that is not the whole image, through the drop-in API. Run it under the knobs that move the road's
decisions -- they are read once per process:
    python tools/fuzz/fuzz_host_road.py [seed] [n]                        the mapped road, 4 MB chunks
    SMOL_HIP_MAPPED_CHUNK_KB=64 python tools/fuzz/fuzz_host_road.py ...   many chunks per image (chunk seams on every few rows)
    SMOL_HIP_BOUNCE_KB=64 python tools/fuzz/fuzz_host_road.py ...         ring slots smaller than most chunks (rows halved; DMA road when a row does not fit)
    SMOL_HIP_MAPPED=0 python tools/fuzz/fuzz_host_road.py ...             the DMA road"""
import os
import random
import sys
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import helpers as H
import smolscale_amd as sm
from test_oracle_vs_reference import random_case

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = random.Random(seed)
bad = 0
kinds = {}
t0 = time.time()
for it in range(N):
    t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 3_000_000 if it % 7 else 12_000_000)
    raw = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
    first = rng.randint(0, h_out - 1) if it % 3 == 0 else 0
    n = rng.randint(1, h_out - first) if it % 3 == 0 else h_out
    want = H.oracle_scale(raw, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    pin_src, pin_dst = rng.choice([0, 1, 2]), rng.choice([0, 1, 2])       # 0 pageable, 1 allocated page-locked, 2 registered in place
    mis = rng.choice([0, 0, 1, 2, 3])
    keep = []
    if pin_src:
        a = sm.PinnedArray(raw.size + 8, register=pin_src == 2)
        keep.append(a)
        sbuf = a.array
    else:
        sbuf = np.zeros(raw.size + 8, np.uint8)
    sbuf[mis:mis + raw.size] = raw
    if pin_dst:
        b = sm.PinnedArray(n * sout, register=pin_dst == 2)
        keep.append(b)
        dst = b.array
    else:
        dst = np.zeros(n * sout, np.uint8)
    dst[:] = 0xA5
    ctx = sm.SmolScaleCtx(sbuf[mis:], t_in, w_in, h_in, sin, None, t_out, w_out, h_out, sout, srgb)
    ctx.batch_full(dst, first, n)
    ctx.destroy()
    rb = w_out * H.bpp(t_out)
    ok = bool(np.array_equal(H.rows_view(dst, n, rb, sout), H.rows_view(want, n, rb, sout)))
    if ok and sout > rb and n > 1:
        gaps = np.lib.stride_tricks.as_strided(dst[rb:], (n - 1, sout - rb), (sout, 1))
        ok = bool(np.all(gaps == 0xA5))
    for p in keep:
        p.free()
    kinds[(pin_src > 0, pin_dst > 0)] = kinds.get((pin_src > 0, pin_dst > 0), 0) + 1
    if not ok:
        bad += 1
        print("BAD", (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n, pin_src, pin_dst, mis), flush=True)
print("host road: cases %d bad %d (source page-locked, destination page-locked): %s  %s  %.0f s"
      % (N, bad, kinds, {k: os.environ[k] for k in ("SMOL_HIP_MAPPED", "SMOL_HIP_MAPPED_CHUNK_KB", "SMOL_HIP_BOUNCE_KB") if k in os.environ},
         time.time() - t0))
sys.exit(1 if bad else 0)
