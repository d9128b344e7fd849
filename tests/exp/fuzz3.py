import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(7)
bad = 0; kc = {}
for it in range(120):
    hx = random.randrange(3); hy = random.randrange(3)
    w_out = random.choice([1, 2, 3, 4, 5, 8, 64, 65, 130, 256, 300, 515]); h_out = random.randint(1, 9)
    if hx == 0: w_out += w_out & 1 if random.random() < 0.7 else 0
    w_in = w_out << (hx + 1); h_in = h_out << (hy + 1)
    t_in = random.randrange(4); t_out = random.randrange(4)
    sin = w_in * 4 + random.choice([0, 16, 32]); sout = w_out * 4 + random.choice([0, 8, 16])
    src = make_input(t_in, w_in, h_in, seed=it, valid=random.random() < 0.5, stride=sin, pad=16)
    # align the source base to 16 bytes
    raw = np.zeros(src.size + 32, np.uint8); off = (-raw.ctypes.data) % 16; s16 = raw[off:off + src.size]; s16[:] = src
    first = random.randint(0, h_out - 1); n = random.randint(1, h_out - first)
    a = oracle_scale(s16, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
    b, k = emu_scale(s16, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
    kc[k] = kc.get(k, 0) + 1
    if not np.array_equal(a, b):
        bad += 1; print("MISMATCH", hx, hy, w_out, h_out, t_in, t_out, k)
print("bad", bad, kc)
