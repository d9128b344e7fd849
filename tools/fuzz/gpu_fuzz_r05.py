"""Randomized GPU parity run over what round 5 built, against the oracle (host- and device-pointer entries, odd strides, row strips):
  gamma   gamma-correct jobs of every type pair -- minified, magnified and mixed axes, boxes -- with 24-bit sources (k_lind's
          three-channel flavour, quad dword stores) in half of them
  hbox    a bilinear / COPY / ONE horizontal axis under a vertical box of ratio 8.5 .. 3000 (k_linh with the box sums in registers),
          gamma-correct, premul16, premul16 + sRGB (19-bit channels) and the class's compressed storage
  rows    row-at-a-time callers through the public API (read-ahead bands), the source rewritten between calls in a third of them
    python tools/fuzz/gpu_fuzz_r05.py [seed] [n_cases]"""
import sys, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import helpers as H
import smolscale_amd as sm

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = random.Random(seed)
bad = 0; fam_count = {}; kernels = {}
t0 = time.time()


def axis(kind, d_out, lo=1.01, hi=7.99):
    if kind == "one": return 1
    if kind == "copy": return d_out
    if kind == "up": return max(2, int(d_out / rng.uniform(1.05, 9)))
    if kind == "down": return min(65535, max(d_out + 1, int(d_out * rng.uniform(lo, hi))))
    return min(65535, int(d_out * rng.uniform(8.5, 40)) + 1)          # box


for it in range(N):
    fam = rng.choice(["gamma", "gamma", "hbox", "hbox", "rows"])
    fam_count[fam] = fam_count.get(fam, 0) + 1
    if fam == "gamma":
        w_out = rng.choice([1, 3, 7, 40, 64, 65, 127, 130, 256, 300, 515, 1000]); h_out = rng.choice([1, 2, 5, 17, 33, 70, 150])
        kh, kv = rng.choice(["down", "down", "up", "copy", "box", "one"]), rng.choice(["down", "down", "up", "copy", "box", "one"])
        w_in, h_in = axis(kh, w_out), axis(kv, h_out)
        if (kh == "one" and w_out == 1) or (kv == "one" and h_out == 1) or w_in * h_in > 1_500_000:
            continue
        t_in = rng.choice([8, 9]) if it & 2 else rng.randrange(10)
        t_out, srgb = rng.randrange(10), 1
    elif fam == "hbox":
        w_out = rng.choice([1, 3, 40, 64, 65, 100, 130, 200, 300]); h_out = rng.choice([1, 2, 3, 5, 9, 20])
        kh = rng.choice(["down", "up", "copy", "one", "down"])
        w_in = axis(kh, w_out)
        if kh == "one" and w_out == 1:
            continue
        ratio = rng.choice([8.5, 9, 13, 20, 40, 64, 65, 100, 255, 256, 300, 1000, 3000])
        h_in = min(65535, int(h_out * ratio * rng.uniform(1.0, 1.3)) + rng.randint(0, 3))
        if w_in * h_in > 4_000_000 or h_in <= h_out * 8:
            continue
        mode = rng.choice(["srgb", "p16", "p16l", "flat"])
        if mode == "srgb": srgb, t_in, t_out = 1, rng.randrange(10), rng.randrange(10)
        elif mode == "p16": srgb, t_in, t_out = 0, rng.choice([4, 5, 6, 7]), rng.choice([4, 5, 6, 7])
        elif mode == "p16l": srgb, t_in, t_out = 1, rng.choice([4, 5, 6, 7]), rng.choice([4, 5, 6, 7])
        else: srgb, t_in, t_out = 0, rng.choice([0, 1, 2, 3, 8, 9]), rng.choice([0, 1, 2, 3, 8, 9])
    else:
        w_in, h_in = rng.randint(8, 700), rng.randint(8, 500)
        w_out, h_out = rng.randint(4, 700), rng.randint(4, 400)
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.choice([0, 0, 1])
    sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 4, 16, 1, 3]); sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4, 16])
    src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
    kn = sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb))
    kernels[kn] = kernels.get(kn, 0) + 1
    rb = w_out * H.bpp(t_out)
    if fam == "rows":
        # rows 1-3 at a time through the public API; every third case rewrites source rows between the calls
        per = rng.choice([1, 1, 2, 3])
        dst = np.zeros(h_out * sout, np.uint8)
        ctx = sm.SmolScaleCtx(src, t_in, w_in, h_in, sin, dst, t_out, w_out, h_out, sout, srgb)
        ok = True
        r2 = np.random.default_rng(it)
        for y in range(0, h_out, per):
            n = min(per, h_out - y)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, y, n) if it % 3 == 0 or y == 0 else None
            if it % 3 != 0 and y == 0:
                whole = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
            ctx.batch(y, n)
            ref = want if it % 3 == 0 else whole[y * sout:y * sout + n * sout]
            if not np.array_equal(H.rows_view(dst[y * sout:], n, rb, sout), H.rows_view(ref, n, rb, sout)):
                ok = False
            if it % 3 == 0:
                lo = int(r2.integers(0, h_in)); hi = min(h_in, lo + int(r2.integers(1, 6)))
                src[lo * sin:lo * sin + (hi - lo - 1) * sin + w_in * H.bpp(t_in)] ^= 0x55
        ctx.destroy()
        if not ok:
            bad += 1
            print("MISMATCH it=%d rows" % it, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, per), flush=True)
        continue
    first = rng.randint(0, h_out - 1) if it % 3 == 0 else 0
    n = rng.randint(1, h_out - first) if it % 3 == 0 else h_out
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    if it & 1:
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n,
                                 src_misalign=rng.choice([0, 0, 1, 2, 4, 8]), dst_misalign=rng.choice([0, 0, 1, 3, 4]))
    else:
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
    if not np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want, n, rb, sout)):
        bad += 1
        print("MISMATCH it=%d" % it, fam, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n), "dev" if it & 1 else "host", kn, flush=True)
print("cases", N, "bad", bad, fam_count, "%.0f s" % (time.time() - t0))
print("kernel histogram:", ", ".join("%s %d (%.1f %%)" % (k, v, 100.0 * v / N) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1])))
