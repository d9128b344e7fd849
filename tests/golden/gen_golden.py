#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (oracle/_ref, generic backend).

Runs only in the build container (needs oracle/_ref, built by `make -C oracle ref`
from /root/reference). Inputs are never stored: they are regenerated from a seed
by the repo's own generator (smol_oracle_fill, xorshift64) -- the fixtures hold
parameters + expected output bytes / digests only.

Writes
  tests/golden/matrix.npz   every pixel-type pair x {srgb 0,1} x filter-size classes
                            on small frames (full output bytes)
  tests/golden/baseline.npz BASELINE.json configs: C1 full output; C2..C5 + the BOX
                            row: FNV-1a-64 digests + a 64x64 crop
  tests/golden/verify.npz   the reference's bytes for the inputs of verify.c's
                            suites (incl. the known-bad pre/unmul bytes, pinned)
  tests/golden/planner.npz  planner tuples from ref_probe_plan() for ~300 dims

Usage: python tests/golden/gen_golden.py
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (w_in, h_in, w_out, h_out): one per reachable filter-class combination
SIZE_CLASSES = [
    (1, 5, 7, 5),        # ONE / COPY
    (9, 1, 9, 4),        # COPY / ONE
    (12, 10, 17, 23),    # 0H magnify / 0H magnify
    (20, 14, 13, 9),     # 0H minify
    (31, 22, 10, 7),     # 1H
    (45, 38, 7, 6),      # 2H
    (64, 32, 16, 8),     # exact 4:1 (1H, F = 128)
    (64, 48, 32, 6),     # exact 2:1 x 8:1
    (100, 9, 11, 9),     # BOX-64 / COPY
    (9, 120, 3, 5),      # 1H / BOX-64
    (300, 40, 1, 4),     # BOX-128 (ratio > 255) / BOX-64
    (2, 2, 1, 1),        # tiny
    (3, 700, 3, 2),      # COPY / BOX-128
    (8200, 2, 1, 2),     # ratio > 8191: sRGB forced off
]


def strided_rows(buf, h, row_bytes, stride):
    return np.lib.stride_tricks.as_strided(buf, (h, row_bytes), (stride, 1))


def gen_matrix():
    cases, blobs = [], []
    seed = 1000
    for (w_in, h_in, w_out, h_out) in SIZE_CLASSES:
        for t_in in range(10):
            for t_out in range(10):
                for srgb in (0, 1):
                    seed += 1
                    valid = seed & 1
                    src = H.make_input(t_in, w_in, h_in, seed=seed, valid=bool(valid), pad=16)
                    out = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
                    cases.append((t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid))
                    blobs.append(out.copy())
    offs = np.cumsum([0] + [b.size for b in blobs]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "matrix.npz"),
                        cases=np.array(cases, np.int64), offsets=offs, blob=np.concatenate(blobs))
    print("matrix: %d cases, %d bytes" % (len(cases), offs[-1]))


def crop64(buf, w, h, bpp, stride):
    cw, ch = min(64, w), min(64, h)
    x0, y0 = (w - cw) // 2, (h - ch) // 2
    rows = strided_rows(buf, h, w * bpp, stride)
    return rows[y0:y0 + ch, x0 * bpp:(x0 + cw) * bpp].copy()


BASELINE_CASES = {
    # name: (t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in)
    "C1": (H.RGBA8_P, 640, 480, H.RGBA8_P, 320, 240, 0, 1, None),
    "C2": (H.RGBA8_P, 4096, 4096, H.RGBA8_P, 1024, 1024, 0, 1, None),
    "C3": (H.RGBA8_U, 1920, 1080, H.RGBA8_P, 7680, 4320, 0, 0, None),
    "C4": (H.RGB8, 65535, 4096, H.RGBA8_P, 8192, 4096, 0, 0, 65535 * 3),
    "C5": (H.RGBA8_P, 256, 256, H.RGBA8_P, 64, 64, 0, 1, None),
    "BOX": (H.RGBA8_P, 4096, 4096, H.RGBA8_P, 448, 448, 0, 1, None),
}


def gen_baseline():
    out = {}
    for name, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in) in BASELINE_CASES.items():
        src = H.make_input(t_in, w_in, h_in, seed=0, valid=bool(valid), stride=stride_in, pad=16)
        dst = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, stride_in=stride_in)
        bo = H.bpp(t_out)
        out[name + "_params"] = np.array([t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid,
                                          stride_in or w_in * H.bpp(t_in)], np.int64)
        out[name + "_fnv"] = np.array([H.fnv1a(dst, w_out * bo, h_out, w_out * bo)], np.uint64)
        out[name + "_crop"] = crop64(dst, w_out, h_out, bo, w_out * bo)
        if name == "C1":
            out[name + "_full"] = dst.copy()
        print(name, "fnv %016x" % int(out[name + "_fnv"][0]))
        del src, dst
    np.savez_compressed(os.path.join(OUT, "baseline.npz"), **out)


# ---- verify.c suites (inputs restated from verify.c:57-98, 227-301, 343-395, 463-514) ----

def ordering_pattern(t, n_pixels):
    """verify.c:57-98: r=0x20+m, g=0x60+m, b=0xa0+m, a=0xff, m cycling 16 steps of 4."""
    pos = {0: (0, 1, 2, 3), 1: (2, 1, 0, 3), 2: (1, 2, 3, 0), 3: (3, 2, 1, 0)}
    bpp = H.bpp(t)
    buf = np.zeros(n_pixels * bpp, np.uint8)
    m = (np.arange(n_pixels) % 16) * 4
    if bpp == 4:
        r, g, b, a = pos[t % 4]
        v = buf.reshape(-1, 4)
        v[:, r] = 0x20 + m
        v[:, g] = 0x60 + m
        v[:, b] = 0xa0 + m
        v[:, a] = 0xff
    else:
        v = buf.reshape(-1, 3)
        order = (0, 1, 2) if t == H.RGB8 else (2, 1, 0)
        v[:, order[0]] = 0x20 + m
        v[:, order[1]] = 0x60 + m
        v[:, order[2]] = 0xa0 + m
    return buf


def gen_verify():
    out = {}
    # ordering: 16384 -> 16383 strip, horizontal and vertical, every type pair (first 64 bytes kept)
    heads = []
    for t_in in range(10):
        for t_out in range(10):
            for vertical in (0, 1):
                src = np.concatenate([ordering_pattern(t_in, 16384), np.zeros(16, np.uint8)])
                if vertical:
                    dst = H.ref_scale(src, t_in, 1, 16384, t_out, 1, 16383, 0)
                else:
                    dst = H.ref_scale(src, t_in, 16384, 1, t_out, 16383, 1, 0)
                heads.append(dst[:64].copy())
    out["ordering_heads"] = np.stack(heads)

    # unassociated alpha: the two 256-case loops of verify.c:227-301
    ua = []
    for i in range(256):
        src = np.array([i, 0xff, 0xff, 0xff, 0, 0, 0, 0] + [0] * 16, np.uint8)
        ua.append(H.ref_scale(src, H.ARGB8_U, 2, 1, H.ARGB8_U, 1, 1, 0)[:4].copy())
    for i in range(256):
        src = np.array([0xff, 0xff, 0xff, 0xff, i, 0, 0, 0] + [0] * 16, np.uint8)
        ua.append(H.ref_scale(src, H.ARGB8_U, 2, 1, H.ARGB8_U, 1, 1, 0)[:4].copy())
    out["unassoc_alpha"] = np.stack(ua)

    # pre/unmul (verify.c:463-514): premul ARGB (i,i,i,i) -> unassociated. The
    # reference's own verify FAILS this suite; its bytes are pinned as they are.
    # Small sizes: full bytes for every alpha; verify.c's own sizes: digests for
    # every 16th alpha, horizontal and vertical.
    pu = []
    for i in range(1, 256):
        for (w_in, w_out) in ((1, 3), (2, 5), (9, 4), (40, 3), (300, 1)):
            for srgb in (0, 1):
                src = np.full(w_in * 4 + 16, i, np.uint8)
                dst = H.ref_scale(src, H.ARGB8_P, w_in, 1, H.ARGB8_U, w_out, 1, srgb)
                pu.append(dst[:4].copy())
    out["preunmul"] = np.stack(pu)
    big = []
    for i in list(range(1, 256, 16)) + [255]:
        for (n_in, n_out) in ((1, 65535), (2, 65535), (65534, 65535), (65535, 1), (65535, 65534)):
            for vertical in (0, 1):
                for srgb in (0, 1):
                    src = np.full(n_in * 4 + 16, i, np.uint8)
                    if vertical:
                        dst = H.ref_scale(src, H.ARGB8_P, 1, n_in, H.ARGB8_U, 1, n_out, srgb)
                    else:
                        dst = H.ref_scale(src, H.ARGB8_P, n_in, 1, H.ARGB8_U, n_out, 1, srgb)
                    big.append((i, n_in, n_out, vertical, srgb, H.fnv1a(dst, n_out * 4, 1, n_out * 4)))
    out["preunmul_big"] = np.array(big, np.uint64)
    np.savez_compressed(os.path.join(OUT, "verify.npz"), **out)
    print("verify: ordering %d, unassoc %d, preunmul %d + %d digests" % (len(heads), len(ua), len(pu), len(big)))


def gen_planner():
    dims = set()
    outs = [1, 2, 3, 7, 8, 100, 255, 1024, 8192, 65535]
    for o in outs:
        for i in (1, 2, o, o + 1, 2 * o - 1, 2 * o, 2 * o + 1, 4 * o, 4 * o + 1, 8 * o, 8 * o + 1,
                  255 * o, 255 * o + 1, 8191 * o, 8191 * o + 1, 65535, 640, 4096, 1920):
            if 1 <= i <= 65535:
                dims.add((i, o))
    dims = sorted(dims)
    rows = []
    lib = H.ref()
    for (i, o) in dims:
        for (t_in, t_out, srgb) in ((0, 0, 0), (0, 0, 1), (4, 5, 0), (4, 0, 1), (8, 9, 1)):
            p = H.RefProbePlan()
            lib.ref_probe_plan(t_in, i, 3, t_out, o, 3, srgb, ctypes.byref(p))     # horizontal axis
            q = H.RefProbePlan()
            lib.ref_probe_plan(t_in, 3, i, t_out, 3, o, srgb, ctypes.byref(q))     # vertical axis
            rows.append((i, o, t_in, t_out, srgb,
                         p.filter_h, p.storage_type, p.gamma_type, p.width_bilin_out, p.width_halvings,
                         p.span_mul_x if p.filter_h == 9 else 0,
                         q.filter_v, q.storage_type, q.gamma_type, q.height_bilin_out, q.height_halvings,
                         q.span_mul_y if q.filter_v == 9 else 0))
    np.savez_compressed(os.path.join(OUT, "planner.npz"), rows=np.array(rows, np.int64))
    print("planner: %d tuples" % len(rows))


def main():
    if not H.have_ref():
        sys.exit("oracle/_ref missing: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    gen_planner()
    gen_matrix()
    gen_verify()
    gen_baseline()


if __name__ == "__main__":
    main()
