"""Kernel LOGIC check without a GPU: the device code of smol_kernels.hip.h compiled
for the host (tests/emu) and run block-by-block, against the oracle. This is a
debugging aid for the build container -- the parity tests proper are the -m gpu
ones, which run the real HIP binary."""
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
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        seen.add(k)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb, first, n, k)
    assert {0, 1} <= seen


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
        src = H.aligned_copy(H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16))
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        got, k = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first, n)
        hits += (k == 2)
        assert np.array_equal(got, want), (hx, hy, w_out, h_out, t_in, t_out, k)
    assert hits >= 40


def test_emulated_kernels_match_golden_subset():
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid), want in C.matrix_cases(stride=23):
        src = H.make_input(t_in, w_in, h_in, seed=seed, valid=valid, pad=16)
        got, _ = H.emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
