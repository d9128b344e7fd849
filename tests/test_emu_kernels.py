"""Kernel LOGIC check without a GPU: the device code of smol_kernels.hip.h compiled
for the host (tests/emu) and run block-by-block, against the oracle. This is a
debugging aid for the build container -- the parity tests proper are the -m gpu
ones, which run the real HIP binary."""
import os
import random

import numpy as np

import cases as C
import helpers as H
from test_oracle_vs_reference import random_case


def test_emulated_general_kernels_match_oracle():
    rng = random.Random(123)
    seen = set()
    for it in range(120):
        while True:
            t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 120_000)
            if w_out * h_out <= 6000:
                break
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        # force_general=1 keeps the per-pixel backstop kernels under test now that
        # k_tile takes every 64bpp-class job
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n,
                             force_general=it & 1)
        seen.add(k)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb, first, n, k)
    assert {0, 1, 3} <= seen


def test_emulated_pow2_kernel_matches_oracle():
    rng = random.Random(7)
    hits = 0
    for it in range(60):
        hx, hy = rng.randrange(3), rng.randrange(3)
        w_out = rng.choice([2, 4, 6, 8, 64, 66, 130, 256, 300, 514])
        h_out = rng.randint(1, 7)
        w_in, h_in = w_out << (hx + 1), h_out << (hy + 1)
        t_in, t_out = rng.randrange(4), rng.randrange(4)
        sin = w_in * 4 + rng.choice([0, 16, 32])
        sout = w_out * 4 + rng.choice([0, 8, 16])
        # 16-byte aligned rows, and (every third case) a base 4 / 8 / 12 bytes past a 16-byte boundary with a stride
        # of any multiple of 4: a crop of a larger image -- the streaming kernel takes both
        mis = rng.choice([4, 8, 12]) if it % 3 == 2 else 0
        if mis:
            sin += rng.choice([0, 4, 8, 12])
        raw = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        buf = H.aligned_copy(np.concatenate([np.zeros(mis, np.uint8), raw]))
        src = buf[mis:]
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        hits += (k == 2)
        assert np.array_equal(got, want), (hx, hy, w_out, h_out, t_in, t_out, k, mis, sin)
    assert hits >= 40


def test_emulated_pow2x_kernel_every_type_pair_and_class():
    """k_pow2x: exact 2:1/4:1/8:1 for all 10x10 type pairs, with and without sRGB
    (64bpp class, premul16, linear), thumbnails narrower than a wave included."""
    rng = random.Random(77)
    hits, kinds = 0, set()
    for it in range(160):
        h = rng.randrange(3)
        w_out = rng.choice([4, 8, 16, 24, 40, 64, 72, 128, 136, 264])
        h_out = rng.randint(1, 36)
        w_in, h_in = w_out << (h + 1), h_out << (h + 1)
        t_in, t_out = (it // 10) % 10, it % 10              # every pair at least once
        srgb = rng.choice([0, 0, 1])
        src = H.aligned_copy(H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), pad=32))
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, None, None, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, None, None, first, n)
        hits += (k == 7)
        kinds.add((H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb).wide, H.bpp(t_in)))
        assert np.array_equal(got, want), (h, w_out, h_out, t_in, t_out, srgb, first, n, k)
    assert hits >= 120 and len(kinds) == 4


def test_emulated_up_kernel_magnification():
    """k_up: bilinear magnification on both axes, every type pair without sRGB (64bpp
    class and premul16), odd strides, row slabs, tiny sources (2 pixels) and ratios
    from 1.01x to 9x."""
    rng = random.Random(808)
    hits = 0
    # the shipped rule sends 64bpp-class magnifications below ~1.7 to k_tile (smol_jobdev.h: up_setup); the emulation
    # library is a -DSMOL_TUNING build, so the whole ratio range can be kept on k_up here
    os.environ["SMOL_UP_ALWAYS"] = "1"
    for it in range(140):
        w_in = rng.choice([2, 3, 5, 8, 17, 64, 100, 257])
        h_in = rng.choice([2, 3, 4, 9, 30, 65])
        w_out = rng.randint(w_in + 1, min(w_in * 9, 1300))
        h_out = rng.randint(h_in + 1, min(h_in * 9, 300))
        t_in, t_out = (it // 10) % 10, it % 10
        # unassociated -> unassociated is the premul16 flavour of k_up
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 16])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4, 16])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        hits += (k == 8)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
    del os.environ["SMOL_UP_ALWAYS"]
    assert hits >= 130


def test_packed_11_bit_tap_equals_the_reference_formula_for_every_input():
    """lerp11_pk (smol_kernel_tile.hip.h: the sRGB-linear class's bilinear tap in packed 16-bit lanes, split weights)
    against q + (((p - q) * F) >> 8) (smolscale-generic.c:1317, :1704) for every 11-bit p, q and every F in 0..256."""
    import ctypes
    lib = H.emu()
    lib.emu_lerp11_exhaustive.restype = ctypes.c_long
    assert lib.emu_lerp11_exhaustive() == 0


def test_emulated_up_kernel_table_driven_flavours():
    """k_up flavours 2 and 3: magnification with with_srgb = 1 (sRGB-linear; premul16 + sRGB),
    every type pair incl. 24-bit, odd strides."""
    rng = random.Random(31337)
    kinds = set()
    for it in range(100):
        w_in = rng.choice([2, 3, 8, 17, 64, 100])
        h_in = rng.choice([2, 3, 9, 30])
        w_out = rng.randint(w_in + 1, min(w_in * 9, 700))
        h_out = rng.randint(h_in + 1, min(h_in * 9, 120))
        t_in, t_out = (it // 10) % 10, it % 10
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 1)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin, sout, first, n)
        if k == 8:
            kinds.add((job.premul16, job.linear))
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
    assert {(0, 1), (1, 1)} <= kinds


def test_emulated_down_kernel_premul16_minification():
    """k_lind: unassociated -> unassociated (premul16) bilinear minification, ratios
    1..8 per axis (0-2 halvings), 1 and 4 columns per lane, row slabs, padded strides."""
    rng = random.Random(4242)
    hits = 0
    for it in range(90):
        w_out = rng.choice([1, 2, 3, 7, 40, 64, 65, 130, 256, 300, 515])
        h_out = rng.choice([1, 2, 5, 17, 33])
        w_in = max(2, min(int(w_out * rng.uniform(1.01, 7.99)) + rng.randint(0, 1), 8 * w_out - 1))
        h_in = max(2, min(int(h_out * rng.uniform(1.01, 7.99)) + rng.randint(0, 1), 8 * h_out - 1))
        if w_in <= w_out or h_in <= h_out:
            continue
        t_in, t_out = rng.choice([4, 5, 6, 7]), rng.choice([4, 5, 6, 7])
        sin = w_in * 4 + rng.choice([0, 0, 4, 16, 12])
        sout = w_out * 4 + rng.choice([0, 0, 1, 4, 16])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=False, stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        hits += (k == 9)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
    assert hits >= 60


def test_emulated_down_kernel_table_driven_flavours():
    """k_lind: sRGB-linear (11-bit channels) and premul16 + sRGB (19-bit),
    every input/output type incl. 24-bit, byte-aligned rows."""
    rng = random.Random(999)
    kinds = set()
    for it in range(150):
        # (from 256 columns and below ratio 2 the 11-bit flavour takes four columns per lane and 16-byte stores)
        w_out = rng.choice([1, 2, 7, 40, 64, 65, 130, 256, 257, 300, 515, 700])
        h_out = rng.choice([1, 2, 5, 17, 33])
        rx = rng.uniform(1.01, 1.99) if it % 3 == 1 else rng.uniform(1.01, 7.99)
        w_in = max(2, min(int(w_out * rx) + rng.randint(0, 1), 8 * w_out - 1))
        h_in = max(2, min(int(h_out * rng.uniform(1.01, 7.99)) + rng.randint(0, 1), 8 * h_out - 1))
        if w_in <= w_out or h_in <= h_out:
            continue
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        srgb = 0 if (t_in in (4, 5, 6, 7) and t_out in (4, 5, 6, 7) and it % 3 == 0) else 1
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 4, 1, 3])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        if not job.wide:
            continue
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        if k == 9:
            kinds.add((job.premul16, job.linear))
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n, k)
    assert {(0, 1), (1, 1)} <= kinds


def test_emulated_two_pass_box_kernels():
    """k_boxh + k_boxv: box on both axes through the intermediate image; 64bpp class at
    ratios >= 12 and every 128bpp-class box job, all types, odd strides, row strips."""
    rng = random.Random(2468)
    seen = set()
    for it in range(70):
        w_out = rng.choice([1, 2, 3, 7, 40, 64, 65, 100])
        h_out = rng.choice([1, 2, 3, 5, 9])
        w_in = min(int(w_out * rng.uniform(8.5, 40)) + 1, 4000)
        h_in = min(int(h_out * rng.uniform(8.5, 40)) + 1, 400)
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.choice([0, 0, 1])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        if job.filter_h != 3 or job.filter_v != 3:
            continue
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 16])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        if k == 10:
            seen.add(job.wide)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n, k)
    assert seen == {0, 1}


def test_emulated_kernels_match_golden_subset():
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid), want in C.matrix_cases(stride=23):
        src = H.make_input(t_in, w_in, h_in, seed=seed, valid=valid, pad=16)
        got, _ = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def _fuzz_tile(rng, dims_fn, n_cases, want_class=3, allow_box=True, max_in=400_000, max_out=8000, dims_h_fn=None):
    import helpers as H
    done = 0
    it = 0
    while done < n_cases:
        it += 1
        w_in, w_out = dims_fn(rng)
        h_in, h_out = (dims_h_fn or dims_fn)(rng)
        if w_in * h_in > max_in or w_out * h_out > max_out:
            continue
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        if 4 <= t_in <= 7 and 4 <= t_out <= 7:
            continue                                   # premul16 -> wide class, not k_tile
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
        if job.wide:
            continue
        done += 1
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 16])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 5, 8])
        raw = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=32)
        src = H.aligned_copy(np.concatenate([np.zeros(rng.choice([0, 0, 1, 2, 4]), np.uint8), raw]))
        src = src[src.size - raw.size:]                # random misalignment of the base
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n, force_general=2)
        assert k == want_class
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, first, n, sin, sout)


def test_emulated_tile_kernel_bilinear_family():
    def dims(rng):
        cls = rng.choice(["one", "copy", "up", "0h", "1h", "2h", "wideup", "widedown"])
        if cls == "one":
            return 1, rng.randint(1, 200)
        if cls == "copy":
            d = rng.randint(1, 300)
            return d, d
        if cls == "up":
            i = rng.randint(2, 60)
            return i, rng.randint(i + 1, 6 * i + 3)
        if cls == "0h":
            o = rng.randint(1, 150)
            return rng.randint(o + 1, 2 * o), o
        if cls == "1h":
            o = rng.randint(1, 100)
            return rng.randint(2 * o + 1, 4 * o), o
        if cls == "2h":
            o = rng.randint(1, 70)
            return rng.randint(4 * o + 1, 8 * o), o
        if cls == "wideup":
            i = rng.randint(40, 120)
            return i, rng.randint(300, 700)
        o = rng.randint(65, 200)
        return rng.randint(o + 1, 8 * o), o
    _fuzz_tile(random.Random(2024), dims, 60)


def test_emulated_tile_kernel_wide_minification_variants():
    """Images wide enough (>= 256 output columns) for the entry points with four columns per lane and
    the vertical mode compiled in: bilinear minification at 0-2 halvings on x, every vertical class on y."""
    def dims_w(rng):
        o = rng.randint(256, 330)
        h = rng.choice([0, 0, 1, 2])
        return rng.randint((o << h) + 1, min(o << (h + 1), 4 * o + 400)), o

    def dims_h(rng):
        cls = rng.choice(["copy", "0h", "1h", "2h", "box", "up"])
        o = rng.randint(1, 14)
        if cls == "copy":
            return o, o
        if cls == "up":
            return rng.randint(2, 6), rng.randint(7, 20)
        if cls == "box":
            return rng.randint(9 * o, 12 * o), o
        h = {"0h": 0, "1h": 1, "2h": 2}[cls]
        return rng.randint((o << h) + 1, o << (h + 1)), o
    _fuzz_tile(random.Random(77), dims_w, 40, max_in=600_000, max_out=8000, dims_h_fn=dims_h)


def test_emulated_tile_kernel_box_filters():
    def dims(rng):
        cls = rng.choice(["box", "box", "boxbig", "copy", "0h", "1h", "up", "one"])
        if cls == "box":
            o = rng.randint(1, 80)
            return rng.randint(8 * o + 1, min(24 * o, 4000)), o
        if cls == "boxbig":
            o = rng.randint(1, 12)
            return rng.randint(24 * o + 1, min(255 * o, 3000)), o
        if cls == "copy":
            d = rng.randint(1, 100)
            return d, d
        if cls == "one":
            return 1, rng.randint(1, 80)
        if cls == "up":
            i = rng.randint(2, 40)
            return i, rng.randint(i + 1, 4 * i)
        if cls == "0h":
            o = rng.randint(1, 80)
            return rng.randint(o + 1, 2 * o), o
        o = rng.randint(1, 60)
        return rng.randint(2 * o + 1, 4 * o), o
    _fuzz_tile(random.Random(77), dims, 60)


def test_emulated_mixed_axis_roads():
    """128bpp-class jobs whose axes differ -- one minified and one magnified, a COPY axis, bilinear x box -- gamma-correct
    (every type pair) and premul16 with and without sRGB, odd strides, row strips. Pairings with a box axis must have
    gone through the two-pass road (k_linh / k_boxh + k_linv / k_boxv, smol_jobdev.h: twopass_setup), the others through
    k_lind with both axes in one kernel (down_setup)."""
    rng = random.Random(1357)
    seen = set()

    def axis(kind, d_out):
        if kind == "copy":
            return d_out
        if kind == "up":
            return max(2, int(d_out / rng.uniform(1.05, 9)))
        if kind == "down":
            return max(d_out + 1, int(d_out * rng.uniform(1.01, 7.99)))
        return int(d_out * rng.uniform(8.5, 30)) + 1

    pairs = [("down", "up"), ("up", "down"), ("copy", "down"), ("down", "copy"), ("copy", "up"), ("up", "copy"),
             ("box", "down"), ("box", "up"), ("box", "copy"), ("down", "box"), ("up", "box"), ("copy", "box")]
    for it in range(144):
        kh, kv = pairs[it % len(pairs)]
        w_out, h_out = rng.choice([2, 3, 7, 40, 65, 130]), rng.choice([2, 5, 17, 33])
        w_in, h_in = axis(kh, w_out), axis(kv, h_out)
        if w_in < 2 or h_in < 2 or (kh != "copy" and w_in == w_out) or (kv != "copy" and h_in == h_out):
            continue
        srgb = 0 if it % 4 == 3 else 1
        if srgb:
            t_in, t_out = rng.randrange(10), rng.randrange(10)
        else:
            t_in, t_out = rng.choice([4, 5, 6, 7]), rng.choice([4, 5, 6, 7])
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 4, 16]) * (1 if H.bpp(t_in) == 4 else rng.choice([1, 1, 3]))
        sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 4])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        assert np.array_equal(got, want), (kh, kv, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n, k)
        if k == (10 if "box" in (kh, kv) else 9):
            seen.add((kh, kv, int(job.premul16)))
    assert {(a, b) for (a, b, _) in seen} == set(pairs)
    assert {p for (a, b, p) in seen if "box" in (a, b)} == {0, 1}
    assert {p for (a, b, p) in seen if "box" not in (a, b)} == {0, 1}


def test_emulated_mixed_axis_road_with_compressed_storage():
    """A box of 256+ pixels on one axis puts a job without sRGB and without unassociated alpha into the 128bpp class's
    compressed storage (8-bit values in 32-bit lanes); with a bilinear, COPY or ONE other axis it takes the two-pass road
    too (k_boxh + k_linv, k_linh + k_boxv), 64 or 16 lanes per column for the longest boxes."""
    rng = random.Random(97)

    def axis(kind, d_out):
        if kind == "one":
            return 1
        if kind == "copy":
            return d_out
        if kind == "up":
            return max(2, int(d_out / rng.uniform(1.05, 9)))
        if kind == "down":
            return max(d_out + 1, int(d_out * rng.uniform(1.01, 7.99)))
        return min(65535, int(d_out * rng.uniform(256, 700)) + 1)

    pairs = [("bigbox", "down"), ("bigbox", "up"), ("bigbox", "copy"), ("down", "bigbox"), ("up", "bigbox"), ("copy", "bigbox"),
             ("bigbox", "one"), ("one", "bigbox")]
    seen = set()
    for it in range(40):
        kh, kv = pairs[it % len(pairs)]
        w_out = rng.choice([2, 3, 5]) if kh == "bigbox" else rng.choice([2, 7, 40, 65])
        h_out = rng.choice([2, 3]) if kv == "bigbox" else rng.choice([2, 5, 17])
        w_in, h_in = axis(kh, w_out), axis(kv, h_out)
        t_in, t_out = rng.choice([0, 1, 2, 3, 8, 9]), rng.randrange(10)
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 4, 16])
        sout = w_out * H.bpp(t_out) + rng.choice([0, 1, 4])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        assert np.array_equal(got, want), (kh, kv, t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
        if k == 10:
            seen.add((kh, kv))
    assert seen == set(pairs)


def test_dispatcher_keeps_proportional_resizes_off_the_backstops():
    """smol_hip_kernel_name(): no size of the reference's proportional benchmark matrix (bench.sh:39-71) -- as it is run,
    with sRGB, with 24-bit and with unassociated types -- ends in k_general / k_wavepx / k_wtile, bar the one 1:1 copy of
    the gamma-correct class; jobs with mixed axes in the 128bpp class take k_lind, with a box axis the two-pass road."""
    import smolscale_amd as sm

    def steps_of(w_in, h_in, smin, smax, n):
        w0, w1, h0, h1 = int(w_in * smin), int(w_in * smax), int(h_in * smin), int(h_in * smax)
        for s in range(n):
            yield (min(max(int(w0 + s * (w1 - w0) / (n - 1.0)), 1), 65535), min(max(int(h0 + s * (h1 - h0) / (n - 1.0)), 1), 65535))

    backstops = ("k_general", "k_wavepx", "k_wtile")
    for (t_in, t_out, srgb) in ((2, 2, 0), (2, 2, 1), (8, 8, 0), (8, 8, 1), (4, 0, 0), (4, 4, 0), (4, 4, 1)):
        for (w, h, smax) in ((1920, 1080, 2.0), (7680, 4320, 1.0)):
            bad = [(wo, ho) for (wo, ho) in steps_of(w, h, 0.01, smax, 200)
                   if (wo, ho) != (w, h) and sm.kernel_name(sm.plan_job(t_in, w, h, t_out, wo, ho, srgb)).startswith(backstops)]
            assert not bad, (t_in, t_out, srgb, w, h, bad[:3])
    assert sm.kernel_name(sm.plan_job(0, 4096, 2048, 0, 2731, 3000, 1)) == "k_lind_batch"        # H down, V up: one kernel
    assert sm.kernel_name(sm.plan_job(4, 1000, 3000, 4, 2731, 3000, 0)) == "k_lind_batch"        # H up, V COPY, premul16
    assert sm.kernel_name(sm.plan_job(0, 4096, 4096, 0, 300, 2000, 1)) == "k_boxh_batch + k_linv_batch"
    assert sm.kernel_name(sm.plan_job(4, 300, 4000, 4, 900, 200, 0)) == "k_linh_batch (vertical box sums in registers) + k_boxv_batch"
    assert sm.kernel_name(sm.plan_job(4, 300, 4000, 4, 900, 200, 1)).startswith("k_linh_batch (vertical box sums")   # premul16 + sRGB: no backstop any more
    assert sm.kernel_name(sm.plan_job(0, 4096, 4096, 0, 2731, 2731, 1)) == "k_lind_batch"
    assert sm.kernel_name(sm.plan_job(8, 4000, 3000, 8, 1600, 1200, 1)) == "k_lind_batch"


def test_emulated_up_kernel_24_bit_output_as_shipped():
    """k_up with 24-bit output (round 5: twelve bytes per lane) under the shipped rule (from x2.6): every 64bpp-class source
    type into RGB8 / BGR8, full and partial tiles, padded rows, row ranges."""
    rng = random.Random(4242)
    hits = 0
    for it in range(90):
        w_in = rng.choice([3, 17, 64, 100, 130])
        h_in = rng.choice([2, 4, 9, 30])
        mag = rng.uniform(2.65, 8.0)
        w_out, h_out = min(int(w_in * mag) + rng.randint(0, 2), 1100), min(int(h_in * rng.uniform(2.65, 8.0)) + rng.randint(0, 2), 200)
        t_in, t_out = rng.randrange(10), rng.choice([8, 9])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
        if job.wide:
            continue
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 16])
        sout = w_out * 3 + rng.choice([0, 0, 1, 4, 16])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        hits += (k == 8)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
    assert hits >= 60


def test_emulated_tile_kernel_two_columns_per_lane():
    """k_tile with TWO columns per lane (round 5): 24-bit sources behind one horizontal halving into dword-aligned 32-bit rows,
    as shipped; and, through the tuning knob of the emulation build, every bilinear x bilinear job of the 64bpp class into
    32-bit rows -- every source type, 0-2 halvings, vertical magnification, partial tiles, padded rows, row ranges."""
    import os
    rng = random.Random(77)
    seen = 0
    for forced in (0, 1):
        if forced:
            os.environ["SMOL_TILE_CPL"] = "2"
        try:
            for it in range(110):
                w_out = rng.choice([128, 130, 200, 256, 257, 300, 515, 700])
                h_out = rng.choice([1, 2, 5, 17, 33])
                rx = rng.uniform(2.01, 3.99) if not forced else rng.uniform(1.01, 7.99)
                ry = rng.choice([rng.uniform(1.01, 7.99), rng.uniform(0.3, 0.99)])
                w_in = max(2, min(int(w_out * rx) + rng.randint(0, 1), 8 * w_out - 1))
                h_in = max(2, min(int(h_out * ry) + rng.randint(0, 1), 8 * h_out - 1))
                if w_in == w_out or h_in == h_out:
                    continue
                t_in = rng.choice([8, 9]) if not forced else rng.choice([0, 1, 2, 3, 4, 5, 8, 9])
                t_out = rng.choice([0, 1, 2, 3, 4, 5, 6, 7])
                job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                if job.wide or job.filter_h != 2 or job.filter_v != 2:
                    continue
                sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 4, 1, 3])
                sout = w_out * 4 + rng.choice([0, 0, 4, 8])
                src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
                first = rng.randint(0, h_out - 1)
                n = rng.randint(1, h_out - first)
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
                got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
                seen += k == 3
                assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k, forced)
        finally:
            os.environ.pop("SMOL_TILE_CPL", None)
    assert seen >= 100


def test_emulated_tile_kernel_one_column_into_24_bit_rows_behind_one_halving():
    """k_tile into 24-bit rows behind one halving on both axes (round 5): one column per lane and row groups of 3 KB for
    unassociated sources, premultiplied ones from ratio 3 and 24-bit ones from ratio 3.7, four columns below -- both sides of
    each threshold, every source type, partial tiles, padded rows, row ranges."""
    rng = random.Random(78)
    seen = 0
    for it in range(120):
        w_out = rng.choice([256, 257, 300, 515, 700])
        h_out = rng.choice([2, 5, 17, 33])
        r = rng.uniform(2.01, 3.99)
        w_in = max(2, min(int(w_out * r) + rng.randint(0, 1), 4 * w_out - 1))
        h_in = max(2, min(int(h_out * rng.uniform(2.01, 3.99)) + rng.randint(0, 1), 4 * h_out - 1))
        t_in = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
        t_out = rng.choice([8, 9])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
        if job.wide or job.filter_h != 2 or job.filter_v != 2:
            continue
        sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 4, 1, 3] if H.bpp(t_in) == 3 else [0, 4])
        sout = w_out * 3 + rng.choice([0, 1, 3, 4])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=0)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        seen += k == 3
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin, sout, first, n, k)
    assert seen >= 100


WIDE_SPAN_CASES = [
    # (t_in, w_in, h_in, t_out, w_out, h_out, srgb): a horizontal box of more than ~9 600 pixels per column, every vertical filter
    (2, 65535, 9, 0, 1, 23, 0),         # bilinear V, compressed storage
    (8, 20001, 6, 5, 1, 10, 0),         # 24-bit rows
    (6, 20001, 10, 6, 1, 6, 1),         # premul16 + gamma
    (0, 9855, 1, 7, 1, 39, 1),          # one source row
    (6, 11376, 16, 1, 1, 16, 0),        # vertical COPY
    (2, 20001, 11, 6, 1, 1, 1),         # box on both axes
    (9, 65535, 11, 1, 1, 1, 0),         # 24-bit, box on both axes, the longest row there is
    (8, 65535, 5, 9, 2, 3, 1),          # two columns of 32 767 pixels
    (4, 30000, 7, 4, 3, 7, 0),          # just at the limit: three columns of 10 000
]


def test_emulated_boxes_wider_than_the_lds_read_straight_from_memory():
    """Round 6: a column's span beyond ~9 600 pixels does not fit four waves' row buffers; k_boxh then reads its pixels straight from
    memory (boxh_direct_body) instead of leaving the job to k_general<true>. Every vertical filter behind it, packed rows and
    rows with a byte of padding (24-bit pixels then start on every byte phase), whole images and a strip."""
    KC_BOX2 = 10
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in enumerate(WIDE_SPAN_CASES):
        for pad in (0, 1):
            sin = w_in * H.bpp(t_in) + pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=bool(k & 1), stride=sin, pad=0)
            first = 0 if pad == 0 else h_out // 3
            n = h_out - first
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, first=first, n=n)
            got, kc = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, first=first, n=n)
            assert kc == KC_BOX2, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, kc)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb, pad)
