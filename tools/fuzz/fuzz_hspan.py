import sys, random
sys.path.insert(0, "tests")
import numpy as np
import helpers as H
KC_HSPAN = 11
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0; seen = 0
for it in range(n_cases):
    h = rng.choice([0, 1, 2])
    t_in = rng.randrange(10)
    if t_in >= 8 and h == 0:
        h = rng.choice([1, 2])
    t_out = rng.randrange(10)
    w_out = rng.choice([256, 257, 300, 511, 512, 700, 1024, 1025, 1300, 2047])
    w_in = (w_out << (h + 1)) + rng.choice([-2, -1, 0, 1, 2])
    hgt = rng.randrange(1, 21)
    stride_in = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 5, 16])
    stride_out = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4, 7])
    src = H.make_input(t_in, w_in, hgt, seed=it, valid=True, stride=stride_in, pad=16)
    first = rng.randrange(0, hgt); n = rng.randrange(1, hgt - first + 1)
    want = H.oracle_scale(src, t_in, w_in, hgt, t_out, w_out, hgt, 0, stride_in=stride_in, stride_out=stride_out, first=first, n=n)
    got, k = H.emu_scale(src, t_in, w_in, hgt, t_out, w_out, hgt, 0, stride_in=stride_in, stride_out=stride_out, first=first, n=n)
    if k != KC_HSPAN:
        continue
    seen += 1
    g = H.rows_view(got, n, w_out * H.bpp(t_out), stride_out); w = H.rows_view(want, n, w_out * H.bpp(t_out), stride_out)
    if not np.array_equal(g, w):
        bad += 1
        ys, xs = np.nonzero(g != w)
        print("MISMATCH", (t_in, w_in, hgt, t_out, w_out, h, stride_in, stride_out, first, n), "first diff row", ys[0], "byte", xs[0], "count", len(ys))
print("hspan cases", seen, "bad", bad)
