"""Emulation fuzz of k_tile's ring staging (smol_kernel_tile.hip.h, round 6) against the oracle, no GPU: 64bpp-class jobs whose
tiles hold more source rows than the ring (tall tiles, box axes, wide rows that take several DMA instructions per row, 24-bit and
byte-misaligned rows), under every ring depth the kernel knows.
    SMOL_TILE_RING=<2|3|4|6|8|12> python tools/fuzz/fuzz_tile_ring.py [seed n]     (tests/emu is a -DSMOL_TUNING build)
    python tools/fuzz/fuzz_tile_ring.py seed n gpu     the same cases, four times the size, on the GPU (device-pointer entry)"""
import os, sys, random
sys.path.insert(0, "tests")
import numpy as np
import helpers as H

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
rng = random.Random(seed)
KC_TILE = 3
bad = seen = 0
for it in range(N):
    cls = rng.choice(["r15", "r25", "r5", "box", "boxbil", "copyv", "up", "wide"])
    if cls == "r15":   w_in, w_out, h_in, h_out = rng.randint(300, 900), 0, rng.randint(40, 120), 0; rx = ry = rng.uniform(1.05, 1.95)
    elif cls == "r25": w_in, w_out, h_in, h_out = rng.randint(300, 900), 0, rng.randint(60, 160), 0; rx = ry = rng.uniform(2.05, 3.95)
    elif cls == "r5":  w_in, w_out, h_in, h_out = rng.randint(600, 1500), 0, rng.randint(80, 200), 0; rx = ry = rng.uniform(4.05, 7.9)
    elif cls == "box": w_in, w_out, h_in, h_out = rng.randint(700, 2500), 0, rng.randint(100, 300), 0; rx = ry = rng.uniform(8.1, 20)
    elif cls == "boxbil": w_in, w_out, h_in, h_out = rng.randint(700, 2500), 0, rng.randint(60, 120), 0; rx = rng.uniform(8.1, 30); ry = rng.uniform(1.1, 3)
    elif cls == "copyv": w_in, w_out, h_in, h_out = rng.randint(300, 900), 0, rng.randint(20, 60), 0; rx = rng.uniform(1.1, 3.5); ry = 1.0
    elif cls == "up":  w_in, w_out, h_in, h_out = rng.randint(150, 400), 0, rng.randint(20, 50), 0; rx = ry = rng.uniform(0.55, 0.95)
    else:              w_in, w_out, h_in, h_out = rng.randint(1500, 4000), 0, rng.randint(30, 60), 0; rx = rng.uniform(1.0, 1.3); ry = rng.uniform(1.0, 2.5)
    if GPU:
        w_in, h_in = w_in * 2, h_in * 2
    w_out = max(1, int(w_in / rx)); h_out = max(1, int(h_in / ry))
    t_in, t_out = rng.randrange(10), rng.randrange(10)
    if H.bpp(t_in) == 4 and 4 <= t_in < 8 and 4 <= t_out < 8:
        t_out = rng.randrange(4)          # unassociated -> unassociated is the 128bpp class
    sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 16])
    sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 2, 8])
    src = H.make_input(t_in, w_in, h_in, seed=it, valid=rng.random() < 0.5, stride=sin, pad=16)
    first = rng.choice([0, 0, h_out // 3])
    n = h_out - first
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n)
    if GPU:
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n, src_misalign=rng.choice([0, 0, 4, 1]))
        k = KC_TILE
    else:
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n)
    rb = w_out * H.bpp(t_out)
    A, B = H.rows_view(want, n, rb, sout), H.rows_view(got, n, rb, sout)
    seen += k == KC_TILE
    if not np.array_equal(A, B):
        bad += 1
        print("MISMATCH", cls, (t_in, w_in, h_in, t_out, w_out, h_out), sin, sout, first, "class", k, flush=True)
print("ring=%s seed=%d: %d cases, %d through k_tile, %d bad" % (os.environ.get("SMOL_TILE_RING"), seed, N, seen, bad))
sys.exit(1 if bad else 0)
