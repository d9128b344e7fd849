"""Oracle vs the committed golden vectors (captured from the real reference by
tests/golden/gen_golden.py). These run anywhere -- no reference, no GPU needed."""
import ctypes

import numpy as np

import cases as C
import helpers as H


def test_luts_match_golden():
    g = C.load_golden("luts.npz")
    import re
    for path, prefix in ((H.os.path.join(H.ROOT, "oracle", "smol_oracle_luts.h"), "oracle_"),
                         (H.os.path.join(H.ROOT, "smolscale_amd", "csrc", "smol_luts.h"), "smol_")):
        text = open(path).read()
        for name in ("from_srgb", "to_srgb", "inv_p8", "inv_p8l", "inv_p16", "inv_p16l"):
            m = re.search(r"%s%s_lut \[\d+\] =\s*\{(.*?)\};" % (prefix, name), text, re.S)
            vals = np.array([int(v, 0) for v in m.group(1).replace("\n", " ").split(",") if v.strip()], np.int64)
            assert np.array_equal(vals, g[name].astype(np.int64)), (path, name)
    # property stated in the reference: to_srgb[from_srgb[i]] == i
    assert all(int(g["to_srgb"][int(g["from_srgb"][i])]) == i for i in range(256))


def test_oracle_matches_golden_matrix():
    n = 0
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid), want in C.matrix_cases():
        src = H.make_input(t_in, w_in, h_in, seed=seed, valid=valid, pad=16)
        got = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        n += 1
    assert n == 2800


def test_oracle_matches_golden_baseline_configs():
    g = C.load_golden("baseline.npz")
    for name in ("C1", "C2", "C5", "BOX", "C3"):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in = [int(v) for v in g[name + "_params"]]
        src = H.make_input(t_in, w_in, h_in, seed=0, valid=bool(valid), stride=stride_in, pad=16)
        out = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, stride_in=stride_in)
        bo = H.bpp(t_out)
        assert H.fnv1a(out, w_out * bo, h_out, w_out * bo) == int(g[name + "_fnv"][0]), name
        if name == "C1":
            assert np.array_equal(out, g["C1_full"])


def test_oracle_matches_golden_c4_strip():
    """C4 (65535x4096 RGB8 -> 8192x4096 RGBA8, odd rowstride): vertical COPY, so a
    row strip of the oracle is compared with the stored centre crop."""
    g = C.load_golden("baseline.npz")
    t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in = [int(v) for v in g["C4_params"]]
    crop = g["C4_crop"]
    y0 = (h_out - 64) // 2
    x0 = (w_out - 64) // 2
    # only rows y0..y0+63 of the input are needed (COPY): regenerate the whole seeded
    # stream (the generator is sequential) but scale just that strip
    src = H.make_input(t_in, w_in, h_in, seed=0, valid=False, stride=stride_in, pad=16)
    out = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, stride_in=stride_in, first=y0, n=64)
    rows = H.rows_view(out, 64, w_out * 4, w_out * 4)
    assert np.array_equal(rows[:, x0 * 4:(x0 + 64) * 4], crop)


def test_oracle_planner_matches_golden_planner():
    rows = C.load_golden("planner.npz")["rows"]
    for r in rows:
        (i, o, t_in, t_out, srgb, fh, st_h, g_h, bil_w, hv_w, mul_x, fv, st_v, g_v, bil_h, hv_h, mul_y) = [int(v) for v in r]
        p = H.OraclePlan()
        assert H.oracle().smol_oracle_plan(t_in, i, 3, t_out, o, 3, srgb, ctypes.byref(p)) == 0
        assert (p.filter_h, p.storage_type, p.gamma_type, p.width_bilin_out) == (fh, st_h, g_h, bil_w), (i, o)
        if 2 <= fh <= 8:
            assert p.width_halvings == hv_w
        if fh == 9:
            assert p.span_mul_x == mul_x
        q = H.OraclePlan()
        assert H.oracle().smol_oracle_plan(t_in, 3, i, t_out, 3, o, srgb, ctypes.byref(q)) == 0
        assert (q.filter_v, q.storage_type, q.gamma_type, q.height_bilin_out) == (fv, st_v, g_v, bil_h), (i, o)
        if fv == 9:
            assert q.span_mul_y == mul_y


def test_opaque_fast_path_constants():
    """The opaque-pixel fast paths of the sRGB-linear class (smol_kernels.hip.h) rely on
    inv_p8l[255] == 1028 (so (s * 1028) >> 10 == s + (s >> 8)) and on un-premultiplying by 255
    being the identity on bytes."""
    import numpy as np
    import os
    luts = np.load(os.path.join(os.path.dirname(__file__), "golden", "luts.npz"))
    assert int(luts["inv_p8l"][255]) == 1028
    inv255 = int(luts["inv_p8"][255])
    assert all(((v * inv255) >> 13) & 0xff == v for v in range(256))
    assert all(((((x + 1) * 256) - 1) >> 8) & 0xff == x for x in range(256))      # premul8(x, 255) == x


def test_inverse_tables_fit_24_bit_multiplies():
    """smol_kernels.hip.h inv_mul(): (value * inverse) is formed with one 24 x 24-bit multiply whose low 32 bits
    are kept; exact because every table entry is below 2^24 (in fact 2^21) and no shift + mask reaches bit 32."""
    import os
    luts = np.load(os.path.join(os.path.dirname(__file__), "golden", "luts.npz"))
    for name, shift, width in (("inv_p8", 13, 8), ("inv_p8l", 10, 11), ("inv_p16", 16, 8), ("inv_p16l", 19, 11)):
        assert int(luts[name].max()) < (1 << 24), name
        assert shift + width <= 32
    # spot-check the identity on the extremes of every table, for lane values up to 2^24 - 1
    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.integers(0, 1 << 24, 4096, dtype=np.uint64), np.array([0, 1, 255, 2047, (1 << 24) - 1], np.uint64)])
    for name, shift, width in (("inv_p8", 13, 8), ("inv_p8l", 10, 11), ("inv_p16", 16, 8), ("inv_p16l", 19, 11)):
        for inv in np.unique(luts[name]).astype(np.uint64)[[0, 1, -2, -1]]:
            full = (vals * inv) >> np.uint64(shift) & np.uint64((1 << width) - 1)
            low32 = ((vals * inv) & np.uint64(0xFFFFFFFF)) >> np.uint64(shift) & np.uint64((1 << width) - 1)
            assert np.array_equal(full, low32), name
