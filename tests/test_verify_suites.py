"""The reference's own verification suites (verify.c) restated against this repo:
the three suites that pass on the reference are known-answer tests for the oracle
(CPU) and for the HIP path (GPU); the fourth (pre/unmul), which FAILS on the
reference itself, is pinned to the reference's actual bytes via tests/golden."""
import numpy as np
import pytest

import cases as C
import helpers as H


def _run(scale, src, t_in, n_in, t_out, n_out, vertical, srgb=0):
    if vertical:
        return scale(src, t_in, 1, n_in, t_out, 1, n_out, srgb)
    return scale(src, t_in, n_in, 1, t_out, n_out, 1, srgb)


def check_ordering(scale, pairs):
    """verify.c:188-225: channel order through every type pair, +-2."""
    g = C.load_golden("verify.npz")["ordering_heads"]
    for (t_in, t_out) in pairs:
        src = C.ordering_pattern(t_in)
        want = C.ordering_pattern(t_out)[:64].astype(int)
        for vertical in (0, 1):
            out = _run(scale, src, t_in, 16384, t_out, 16383, vertical)
            assert np.all(np.abs(out[:64].astype(int) - want) <= 2), (t_in, t_out, vertical)
            assert np.array_equal(out[:64], g[(t_in * 10 + t_out) * 2 + vertical]), (t_in, t_out, vertical)


def check_unassociated_alpha(scale):
    """verify.c:227-301"""
    g = C.load_golden("verify.npz")["unassoc_alpha"]
    for k, (inp, want, tol) in enumerate(C.unassoc_alpha_cases()):
        src = np.array(inp + [0] * 16, np.uint8)
        out = scale(src, H.ARGB8_U, 2, 1, H.ARGB8_U, 1, 1, 0)
        assert np.all(np.abs(out[:4].astype(int) - np.array(want)) <= tol), (inp, out[:4], want)
        assert np.array_equal(out[:4], g[k])


def check_saturation(scale, pairs, sizes=C.VERIFY_SIZES):
    """verify.c:343-395: all-0xff stays all-0xff."""
    src = np.full(65536 * 4 + 16, 0xff, np.uint8)
    for (t_in, t_out) in pairs:
        for (n_in, n_out) in sizes:
            for vertical in (0, 1):
                for srgb in (0, 1):
                    out = _run(scale, src, t_in, n_in, t_out, n_out, vertical, srgb)
                    assert out.min() == 0xff, (t_in, t_out, n_in, n_out, vertical, srgb)


def check_preunmul_pinned(scale, alphas=None):
    """verify.c:463-514 inputs; expectations are the reference's actual bytes."""
    g = C.load_golden("verify.npz")
    small = g["preunmul"]
    k = 0
    for i in range(1, 256):
        for (w_in, w_out) in ((1, 3), (2, 5), (9, 4), (40, 3), (300, 1)):
            for srgb in (0, 1):
                if alphas is None or i in alphas:
                    src = np.full(w_in * 4 + 16, i, np.uint8)
                    out = scale(src, H.ARGB8_P, w_in, 1, H.ARGB8_U, w_out, 1, srgb)
                    assert np.array_equal(out[:4], small[k]), (i, w_in, w_out, srgb)
                k += 1
    for row in g["preunmul_big"]:
        i, n_in, n_out, vertical, srgb, digest = [int(v) for v in row]
        if alphas is not None and i not in alphas:
            continue
        src = np.full(n_in * 4 + 16, i, np.uint8)
        out = _run(scale, src, H.ARGB8_P, n_in, H.ARGB8_U, n_out, vertical, srgb)
        assert H.fnv1a(out, n_out * 4, 1, n_out * 4) == digest, (i, n_in, n_out, vertical, srgb)


ALL_PAIRS = [(a, b) for a in range(10) for b in range(10)]
SOME_PAIRS = [(0, 0), (1, 6), (2, 9), (3, 4), (4, 5), (5, 2), (6, 7), (7, 8), (8, 1), (9, 3), (8, 8), (9, 8)]


# ---- CPU: the oracle passes the reference's own known-answer tests ----------------

def test_oracle_ordering():
    check_ordering(H.oracle_scale, ALL_PAIRS)


def test_oracle_unassociated_alpha():
    check_unassociated_alpha(H.oracle_scale)


def test_oracle_saturation():
    check_saturation(H.oracle_scale, SOME_PAIRS)


def test_oracle_preunmul_pinned_to_reference_bytes():
    check_preunmul_pinned(H.oracle_scale)


# ---- GPU: the HIP path passes them too ------------------------------------------------

@pytest.mark.gpu
def test_hip_ordering():
    check_ordering(H.hip_scale, ALL_PAIRS)


@pytest.mark.gpu
def test_hip_unassociated_alpha():
    check_unassociated_alpha(H.hip_scale)


@pytest.mark.gpu
def test_hip_saturation():
    check_saturation(H.hip_scale, SOME_PAIRS)


@pytest.mark.gpu
def test_hip_preunmul_pinned_to_reference_bytes():
    check_preunmul_pinned(H.hip_scale, alphas=set(range(1, 256, 16)) | {2, 3, 0x7e, 0x7f, 0x80, 255})
