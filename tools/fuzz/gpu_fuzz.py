"""Long randomized GPU parity run (every kernel class) against the oracle; prints mismatching cases.
    python tools/fuzz/gpu_fuzz.py [seed] [n_cases]"""
import sys, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import helpers as H
from test_oracle_vs_reference import random_case
import smolscale_amd as sm

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = random.Random(seed)
bad = 0; fam_count = {}; kernels = {}
t0 = time.time()
for it in range(N):
    fam = rng.choice(["any", "any", "pow2x", "up", "up", "bigw", "thin", "hspan"])
    srgb = 0
    if fam == "any":
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 600_000)
    elif fam == "pow2x":
        h = rng.randrange(3); w_out = rng.choice([4, 8, 12, 64, 100, 256, 1000]); h_out = rng.randint(1, 90)
        w_in, h_in = w_out << (h + 1), h_out << (h + 1)
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.randrange(2)
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 16, 4, 1]); sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 16, 4, 2])
    elif fam == "hspan":
        # k_hspan: vertical copy, horizontal bilinear at a nearly power-of-two ratio (BASELINE config 4's class)
        h = rng.randrange(3); w_out = rng.choice([256, 257, 300, 511, 1024, 1500, 4097]); h_in = h_out = rng.randint(1, 40)
        w_in = (w_out << (h + 1)) + rng.choice([-2, -1, -1, 0, 1, 1, 2])
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), 0
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 16]); sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4, 16])
    elif fam == "up":
        w_in, h_in = rng.randint(2, 400), rng.randint(2, 200)
        w_out = rng.randint(w_in + 1, min(w_in * 12, 3000)); h_out = rng.randint(h_in + 1, min(h_in * 12, 700))
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.choice([0, 0, 0, 1])
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 16]); sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4, 16])
    elif fam == "bigw":
        w_in = rng.choice([65535, 40000, 20001]); h_in = rng.randint(2, 12)
        w_out = rng.choice([65535, 8192, 30000, 1, 333]); h_out = rng.randint(1, 30)
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.choice([0, 0, 1])
        sin = w_in * H.bpp(t_in) + rng.choice([0, 1]); sout = w_out * H.bpp(t_out)
    else:
        w_in, h_in = rng.randint(1, 6), rng.randint(1, 3000)
        w_out, h_out = rng.randint(1, 9), rng.randint(1, 3000)
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.randrange(2)
        sin = w_in * H.bpp(t_in) + rng.choice([0, 1, 5]); sout = w_out * H.bpp(t_out) + rng.choice([0, 3])
    fam_count[fam] = fam_count.get(fam, 0) + 1
    src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
    first = rng.randint(0, h_out - 1) if it % 3 == 0 else 0
    n = rng.randint(1, h_out - first) if it % 3 == 0 else h_out
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    # which kernel family the dispatcher hands these rows to (packed aligned rows; the run itself also uses odd strides)
    kn = sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb), first_row=first, n_rows=n)
    kernels[kn] = kernels.get(kn, 0) + 1
    if it & 1:
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n,
                                 src_misalign=rng.choice([0, 0, 1, 2, 4, 8]), dst_misalign=rng.choice([0, 0, 1, 3, 4]))
    else:
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    rb = w_out * H.bpp(t_out)
    if not np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want, n, rb, sout)):
        bad += 1
        print("MISMATCH it=%d" % it, fam, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n), "dev" if it & 1 else "host", flush=True)
print("cases", N, "bad", bad, fam_count, "%.0f s" % (time.time() - t0))
print("kernel histogram:", ", ".join("%s %d (%.1f %%)" % (k, v, 100.0 * v / N) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1])))
