import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
exec(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "fuzz1.py")).read().split("bad = 0")[0].split("def rand_dims")[0])
src_code = open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "fuzz1.py")).read()
exec("def rand_dims" + src_code.split("def rand_dims")[1].split("bad = 0")[0])
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
kc = {}
for it in range(N):
    while True:
        w_in, w_out = rand_dims(); h_in, h_out = rand_dims()
        if w_in * h_in <= 400_000 and w_out*h_out <= 20000: break
    t_in = random.randrange(10); t_out = random.randrange(10)
    srgb = random.randrange(2)
    valid = random.random() < 0.5
    sin = w_in * bpp(t_in) + random.choice([0, 0, 1, 3, 4, 7])
    sout = w_out * bpp(t_out) + random.choice([0, 0, 1, 2, 5, 8])
    src = make_input(t_in, w_in, h_in, seed=it, valid=valid, stride=sin, pad=16)
    first = random.randint(0, h_out - 1); n = random.randint(1, h_out - first)
    a = oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    b, k = emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    kc[k] = kc.get(k, 0) + 1
    if not np.array_equal(a, b):   # includes padding bytes staying at fill
        bad += 1
        rb = w_out * bpp(t_out)
        A = rows_view(a, n, rb, sout); B = rows_view(b, n, rb, sout)
        nz = np.argwhere(A != B)
        job = plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        print("MISMATCH", TYPE_NAMES[t_in], (w_in, h_in), "->", TYPE_NAMES[t_out], (w_out, h_out), "srgb", srgb, "rows", first, n,
              "filt", job.filter_h, job.filter_v, "halv", job.halvings_h, job.halvings_v, "wide", job.wide, "lin", job.linear, "p16", job.premul16,
              "k", k, "ndiff", len(nz), "first", nz[0] if len(nz) else "PADDING", (A[tuple(nz[0])], B[tuple(nz[0])]) if len(nz) else "")
        if bad > 10: break
print("cases", it + 1, "bad", bad, "classes", kc)
