#!/usr/bin/env python3
"""Which kernel family takes what (no GPU needed: smol_hip_kernel_name() is the dispatcher's own decision).

  - the reference's proportional benchmark matrix (bench.sh:39-71: 1080p / 4K / 8K x 200 output sizes), as the sweep
    runs it (ARGB8 premultiplied, with_srgb 0) and with with_srgb 1 / with 24-bit and unassociated types;
  - a random population of jobs (every type pair, both gamma settings, ratios 1/64 .. 12 per axis independently).
The correctness backstops -- k_wtile, k_wavepx, k_general -- should take (next to) nothing of either.

    python tools/kernel_histogram.py > profiles/r04_kernel_histogram.txt"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smolscale_amd as sm  # noqa: E402


def steps_of(w_in, h_in, smin, smax, n):
    w0, w1 = int(w_in * smin), int(w_in * smax)
    h0, h1 = int(h_in * smin), int(h_in * smax)
    ws, hs = (w1 - w0) / (n - 1.0), (h1 - h0) / (n - 1.0)
    for s in range(n):
        yield (min(max(int(w0 + s * ws), 1), 65535), min(max(int(h0 + s * hs), 1), 65535))


def show(title, hist):
    n = sum(hist.values())
    print("%-58s %s" % (title, ", ".join("%s %d (%.1f %%)" % (k, v, 100.0 * v / n)
                                         for k, v in sorted(hist.items(), key=lambda kv: -kv[1]))))


def main():
    print("# kernel family per job, by the dispatcher (smol_hip_kernel_name; packed, aligned rows; whole images)")
    for (t_in, t_out, srgb, what) in ((2, 2, 0, "ARGB8 premul (the reference's benchmark)"), (2, 2, 1, "ARGB8 premul, with_srgb"),
                                      (8, 8, 0, "RGB8"), (8, 8, 1, "RGB8, with_srgb"), (4, 0, 0, "rgbA -> RGBA"),
                                      (4, 4, 0, "rgbA -> rgbA (premul16)"), (4, 4, 1, "rgbA -> rgbA, with_srgb")):
        for (w, h, smax) in ((1920, 1080, 2.0), (3840, 2160, 2.0), (7680, 4320, 1.0)):
            hist = {}
            for (w_out, h_out) in steps_of(w, h, 0.01, smax, 200):
                k = sm.kernel_name(sm.plan_job(t_in, w, h, t_out, w_out, h_out, srgb))
                hist[k] = hist.get(k, 0) + 1
            show("sweep %dx%d x 200 sizes, %s:" % (w, h, what), hist)
    rng = random.Random(4)
    for srgb in (0, 1):
        hist = {}
        for _ in range(20000):
            w_in, h_in = rng.randint(2, 6000), rng.randint(2, 4000)
            rx, ry = 2.0 ** rng.uniform(-6, 3.6), 2.0 ** rng.uniform(-6, 3.6)
            w_out, h_out = max(1, min(65535, int(w_in * rx))), max(1, min(65535, int(h_in * ry)))
            k = sm.kernel_name(sm.plan_job(rng.randrange(10), w_in, h_in, rng.randrange(10), w_out, h_out, srgb))
            hist[k] = hist.get(k, 0) + 1
        show("20 000 random jobs (axes scaled independently), with_srgb %d:" % srgb, hist)


if __name__ == "__main__":
    main()
