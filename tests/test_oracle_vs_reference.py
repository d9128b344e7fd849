"""Pins the oracle (oracle/smol_oracle.c) against the REAL reference compiled into
oracle/_ref (generic backend): planner decisions, precalc tables and output bytes.
Skipped only where oracle/_ref has not been built (it is prebuilt and shipped)."""
import ctypes
import random

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (make -C oracle ref)")


def _dims(rng, cls):
    if cls == "one":
        return 1, rng.randint(1, 40)
    if cls == "copy":
        d = rng.randint(1, 60)
        return d, d
    if cls == "up":
        i = rng.randint(2, 40)
        return i, rng.randint(i + 1, 4 * i + 3)
    if cls == "0h":
        o = rng.randint(1, 40)
        return rng.randint(o + 1, 2 * o), o
    if cls == "1h":
        o = rng.randint(1, 30)
        return rng.randint(2 * o + 1, 4 * o), o
    if cls == "2h":
        o = rng.randint(1, 20)
        return rng.randint(4 * o + 1, 8 * o), o
    if cls == "box64":
        o = rng.randint(1, 8)
        return rng.randint(8 * o + 1, min(255 * o, 2000)), o
    if cls == "box128":
        o = rng.randint(1, 3)
        return rng.randint(255 * o + 1, 255 * o + 1200), o
    if cls == "huge":
        return rng.randint(8192, 12000), 1
    return rng.randint(1, 4), rng.randint(1, 4)


CLASSES = ["one", "copy", "up", "0h", "1h", "2h", "box64", "box128", "huge", "tiny"]


def random_case(rng, max_in=1_500_000):
    while True:
        w_in, w_out = _dims(rng, rng.choice(CLASSES))
        h_in, h_out = _dims(rng, rng.choice(CLASSES))
        if w_in * h_in <= max_in:
            break
    t_in, t_out = rng.randrange(10), rng.randrange(10)
    srgb = rng.randrange(2)
    sin = w_in * H.bpp(t_in) + rng.choice([0, 0, 1, 3, 4, 7])
    sout = w_out * H.bpp(t_out) + rng.choice([0, 0, 1, 2, 5, 8])
    return t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout


def test_oracle_bytes_equal_reference_random_sweep():
    rng = random.Random(20260101)
    for it in range(400):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        a = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
        b = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
        rb = w_out * H.bpp(t_out)
        assert np.array_equal(H.rows_view(a, h_out, rb, sout), H.rows_view(b, h_out, rb, sout)), \
            (t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def test_oracle_every_type_pair_every_mode():
    # every in/out pair x srgb on one bilinear and one box geometry
    seed = 0
    for (w_in, h_in, w_out, h_out) in ((37, 29, 11, 40), (120, 30, 5, 2)):
        for t_in in range(10):
            for t_out in range(10):
                for srgb in (0, 1):
                    seed += 1
                    src = H.make_input(t_in, w_in, h_in, seed=seed, valid=False, pad=16)
                    a = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
                    b = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
                    assert np.array_equal(a, b), (t_in, t_out, srgb, w_in, w_out)


def test_oracle_strip_split_equals_reference_one_shot():
    rng = random.Random(5)
    for it in range(25):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 200_000)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=False, stride=sin, pad=16)
        full = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
        rb = w_out * H.bpp(t_out)
        F = H.rows_view(full, h_out, rb, sout)
        y = 0
        while y < h_out:
            n = rng.randint(1, max(1, h_out // 3))
            n = min(n, h_out - y)
            part = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, y, n)
            assert np.array_equal(H.rows_view(part, n, rb, sout), F[y:y + n])
            y += n


def test_oracle_planner_equals_reference_planner():
    rng = random.Random(11)
    lib = H.ref()
    for it in range(3000):
        i = rng.choice([1, 2, 3, rng.randint(1, 70), rng.randint(1, 65535), 65535])
        o = rng.choice([1, 2, rng.randint(1, 70), rng.randint(1, 65535), max(1, i // 8), max(1, i // 255), 65535])
        t_in, t_out, srgb = rng.randrange(10), rng.randrange(10), rng.randrange(2)
        ref = H.RefProbePlan()
        lib.ref_probe_plan(t_in, i, 5, t_out, o, 7, srgb, ctypes.byref(ref))
        orc = H.OraclePlan()
        assert H.oracle().smol_oracle_plan(t_in, i, 5, t_out, o, 7, srgb, ctypes.byref(orc)) == 0
        assert (orc.filter_h, orc.filter_v, orc.storage_type, orc.gamma_type) == \
               (ref.filter_h, ref.filter_v, ref.storage_type, ref.gamma_type), (i, o, t_in, t_out, srgb)
        assert orc.width_bilin_out == ref.width_bilin_out
        if 2 <= ref.filter_h <= 8:
            assert orc.width_halvings == ref.width_halvings
        if ref.filter_h == 9:
            assert orc.span_mul_x == ref.span_mul_x, (i, o)


def test_oracle_precalc_tables_equal_reference_tables():
    rng = random.Random(3)
    buf_a = np.zeros(140000, np.uint16)
    buf_b = np.zeros(140000, np.uint16)
    dims = [(640, 320), (4096, 1024), (1920, 7680), (65535, 8192), (4096, 448), (65535, 1), (65535, 65534),
            (2, 65535), (300, 7), (9, 1), (17, 2)]
    dims += [(rng.randint(2, 5000), rng.randint(1, 5000)) for _ in range(150)]
    for (i, o) in dims:
        for axis in (0, 1):
            args = (i, 3, o, 3) if axis == 0 else (3, i, 3, o)
            na = H.oracle().smol_oracle_precalc(*args, axis, buf_a.ctypes.data, buf_a.size)
            nb = H.ref().ref_probe_precalc(*args, axis, buf_b.ctypes.data, buf_b.size)
            assert na == nb, (i, o, axis)
            assert np.array_equal(buf_a[:na], buf_b[:nb]), (i, o, axis)
        ref = H.RefProbePlan()
        H.ref().ref_probe_plan(0, i, 3, 0, o, 3, 0, ctypes.byref(ref))
        orc = H.OraclePlan()
        H.oracle().smol_oracle_plan(0, i, 3, 0, o, 3, 0, ctypes.byref(orc))
        if ref.filter_h == 9:
            assert orc.span_mul_x == ref.span_mul_x, (i, o)


def test_avx2_reference_agrees_except_known_p16_saturation():
    """AVX2 == generic except unassociated->unassociated (saturate vs wrap)."""
    if not H.have_ref("avx2"):
        pytest.skip("avx2 reference not built")
    rng = random.Random(77)
    for it in range(60):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 200_000)
        if t_in in (4, 5, 6, 7) and t_out in (4, 5, 6, 7):
            continue
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=True, stride=sin, pad=16)
        a = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, kind="generic")
        b = H.ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, kind="avx2")
        rb = w_out * H.bpp(t_out)
        assert np.array_equal(H.rows_view(a, h_out, rb, sout), H.rows_view(b, h_out, rb, sout))
