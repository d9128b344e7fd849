"""The library's CPU backend (smolscale_amd/csrc/smol_cpu.c) against the pinned oracle
and the reference-captured golden vectors: it is product code, so it gets the parity tests a kernel
gets -- all 2 800 matrix goldens, the BASELINE digests, a randomized sweep over every pixel-type pair,
filter class, odd stride and row range, strip splits, concurrency -- and the explicit-selection rules of
smolscale_api.c (SMOL_BACKEND). Runs without a GPU. Reference: smolscale-generic.c:185-2318;
wrap-vs-saturate of the premul16 packer: generic wraps (:290-299), AVX2 saturates (smolscale-avx2.c:1045-1047),
this backend wraps."""
import ctypes
import os
import random
import subprocess
import sys
import threading

import numpy as np
import pytest

import cases as C
import helpers as H
import smolscale_amd as sm
from test_oracle_vs_reference import random_case


def cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0, stride_in=None, stride_out=None,
              first=0, n=None, fill=0xA5):
    stride_in = stride_in or w_in * H.bpp(t_in)
    stride_out = stride_out or w_out * H.bpp(t_out)
    n = h_out - first if n is None else n
    dst = np.full(n * stride_out, fill, np.uint8)
    job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    rc = sm.lib().smol_cpu_scale_rows(ctypes.byref(job), src.ctypes.data, stride_in, dst.ctypes.data, stride_out, first, n)
    assert rc == 0
    return dst


def test_cpu_backend_matches_every_golden_matrix_vector():
    n = 0
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid), want in C.matrix_cases():
        src = H.make_input(t_in, w_in, h_in, seed=seed, valid=valid, pad=16)
        got = cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        n += 1
    assert n == 2800


def test_cpu_backend_baseline_digests():
    g = C.load_golden("baseline.npz")
    for name in ("C1", "C2", "C5", "BOX"):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in = [int(v) for v in g[name + "_params"]]
        src = H.make_input(t_in, w_in, h_in, seed=0, valid=bool(valid), stride=stride_in, pad=16)
        out = cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, stride_in=stride_in)
        bo = H.bpp(t_out)
        assert H.fnv1a(out, w_out * bo, h_out, w_out * bo) == int(g[name + "_fnv"][0]), name


def test_cpu_backend_random_sweep_vs_oracle():
    rng = random.Random(424242)
    for it in range(300):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 400_000)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        first = rng.randint(0, h_out - 1) if it % 3 == 0 else 0
        n = rng.randint(1, h_out - first) if it % 3 == 0 else h_out
        a = cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        b = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        rb = w_out * H.bpp(t_out)
        assert np.array_equal(H.rows_view(a, n, rb, sout), H.rows_view(b, n, rb, sout)), \
            (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        # bytes between rows stay untouched
        if sout > rb and n > 1:
            gaps = np.lib.stride_tricks.as_strided(a[rb:], (n - 1, sout - rb), (sout, 1))
            assert np.all(gaps == 0xA5)


def test_cpu_fast_class_every_channel_order_and_misalignment():
    """The AVX2 / AVX-512 paths (the 64bpp class in packed 16-bit lanes, the 128bpp class in 32-bit lanes; 3- and 4-byte pixels,
    either alpha convention on either side; bilinear, box, copy or one-pixel on either axis, ratios above 255, with and without gamma):
    all 100 type pairs, 0-2 halvings per axis, magnification, widths that are not multiples of the vector width,
    byte-misaligned rows, padded destination rows whose padding must stay untouched (24-bit rows are stored with 16-byte
    stores that overlap the next pixels)."""
    rng = random.Random(7)
    for t_in in range(10):
        for t_out in range(10):
            for (w_in, h_in, w_out, h_out) in ((64, 48, 32, 24), (203, 101, 51, 26), (333, 90, 47, 13), (40, 30, 97, 71), (77, 50, 50, 49),
                                               (400, 300, 40, 30), (333, 200, 13, 7), (400, 60, 41, 40), (60, 400, 40, 39), (2000, 9, 8, 1),
                                               (2100, 530, 8, 2), (64, 48, 64, 24), (67, 48, 31, 48), (1, 48, 10, 24), (64, 1, 32, 7),
                                               (37, 11, 37, 11), (500, 30, 500, 3)):
                for srgb in (0, 1):
                    bi, bo = H.bpp(t_in), H.bpp(t_out)
                    sin = w_in * bi + rng.choice([0, 1, 3])
                    sout = w_out * bo + rng.choice([0, 0, 5, 16])
                    raw = H.make_input(t_in, w_in, h_in, seed=rng.randrange(1000), valid=bool(rng.getrandbits(1)), stride=sin, pad=32)
                    mis = rng.choice([0, 1, 2])
                    buf = np.zeros(raw.size + 8, np.uint8)
                    buf[mis:mis + raw.size] = raw
                    a = cpu_scale(buf[mis:], t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
                    b = H.oracle_scale(raw, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout)
                    rb = w_out * bo
                    assert np.array_equal(H.rows_view(a, h_out, rb, sout), H.rows_view(b, h_out, rb, sout)), \
                        (t_in, t_out, w_in, h_in, w_out, h_out, srgb)
                    if sout > rb:
                        gaps = np.lib.stride_tricks.as_strided(a[rb:], (h_out - 1, sout - rb), (sout, 1))
                        assert np.all(gaps == 0xA5), (t_in, t_out, w_out)


def test_generic_road_without_simd():
    """SMOL_CPU_SIMD=0 (read once per process): what a CPU without AVX2 runs -- the golden vectors, the digests and the random
    sweep once more, down the 32-bit-lane road that the SIMD classes otherwise leave to the last pixels of a row."""
    import subprocess
    import sys
    env = dict(os.environ, SMOL_CPU_SIMD="0")
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                          "golden_matrix or baseline_digests or random_sweep or plan_handle"],
                         env=env, capture_output=True, text=True, timeout=1800, cwd=H.ROOT)
    assert run.returncode == 0 and " passed" in run.stdout, (run.stdout[-2000:], run.stderr[-2000:])


def test_cpu_backend_strips_and_threads():
    """smolscale.h:71-73: concurrent batches on disjoint rows of one image give the one-shot bytes."""
    t_in, w_in, h_in, t_out, w_out, h_out = 5, 400, 300, 2, 130, 97
    src = H.make_input(t_in, w_in, h_in, seed=3, pad=16)
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1)
    out = np.zeros(h_out * w_out * 4, np.uint8)
    job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 1)

    def strip(first, n):
        rc = sm.lib().smol_cpu_scale_rows(ctypes.byref(job), src.ctypes.data, w_in * 4,
                                          out[first * w_out * 4:].ctypes.data, w_out * 4, first, n)
        assert rc == 0
    ts = [threading.Thread(target=strip, args=(f, min(13, h_out - f))) for f in range(0, h_out, 13)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert np.array_equal(out, want)


def _run_api(backend, expect_ok=True):
    """smol_scale_simple() in a child process with SMOL_BACKEND set (the choice is read per context)."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, helpers as H, smolscale_amd as sm\n"
        "src = H.make_input(4, 97, 61, seed=1, pad=16)\n"
        "dst = np.zeros(140 * 233 * 4, np.uint8)\n"
        "sm.smol_scale_simple(src, 4, 97, 61, 97 * 4, dst, 1, 233, 140, 233 * 4, 0)\n"
        "want = H.oracle_scale(src, 4, 97, 61, 1, 233, 140)\n"
        "print('EQUAL' if np.array_equal(dst, want) else 'DIFFERENT')\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ)
    if backend is None:
        env.pop("SMOL_BACKEND", None)
    else:
        env["SMOL_BACKEND"] = backend
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)


def test_public_api_on_the_cpu_backend_when_selected():
    run = _run_api("cpu")
    assert run.returncode == 0 and "EQUAL" in run.stdout, run.stdout + run.stderr


@pytest.mark.skipif(H.have_gpu(), reason="needs a machine WITHOUT a GPU")
def test_default_backend_still_fails_loudly_without_a_gpu_and_auto_says_what_it_does():
    run = _run_api(None)
    assert run.returncode != 0 and "no usable HIP device" in run.stderr       # abort(), not a silent fallback
    run = _run_api("auto")
    assert run.returncode == 0 and "EQUAL" in run.stdout
    assert "using cpu" in run.stderr                                          # never silent
    run = _run_api("bogus")
    assert run.returncode != 0 and "is not one of" in run.stderr


def test_cpu_plan_handle_row_batches_and_hand_built_jobs():
    """smol_cpu_plan_*: the tap tables are built once per plan -- rows asked for one at a time (the reference's own row
    loop, smolscale.c:507-546) through ONE plan from several threads equal the oracle; a hand-built job that the planner
    could not have produced is rejected instead of indexing outside the image."""
    import threading
    lib = sm.lib()
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in ((0, 640, 480, 1, 213, 160, 0), (8, 900, 40, 9, 31, 7, 1),
                                                          (4, 77, 300, 5, 200, 90, 0)):
        src = H.make_input(t_in, w_in, h_in, seed=5, valid=True, pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        plan = ctypes.c_void_p()
        assert lib.smol_cpu_plan_create(ctypes.byref(job), ctypes.byref(plan)) == 0
        got = np.zeros_like(want)
        s_out = w_out * H.bpp(t_out)

        def rows(first, n):
            for y in range(first, first + n):
                rc = lib.smol_cpu_plan_scale_rows(plan, src.ctypes.data, w_in * H.bpp(t_in),
                                                  got.ctypes.data + y * s_out, s_out, y, 1)
                assert rc == 0
        ts = [threading.Thread(target=rows, args=sm.strip_rows(k, 4, h_out)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        lib.smol_cpu_plan_destroy(plan)
    # hand-built jobs: bilinear over one pixel, a filter id / halving count / pixel type out of range, copy with unequal sizes
    good = H.plan_job(0, 64, 64, 0, 32, 32, 0)
    for field, value in (("width_in", 1), ("filter_h", 7), ("halvings_v", 3), ("pixel_type_in", 12)):
        bad = sm.SmolHipJob.from_buffer_copy(bytes(good))
        setattr(bad, field, value)
        plan = ctypes.c_void_p()
        assert lib.smol_cpu_plan_create(ctypes.byref(bad), ctypes.byref(plan)) < 0, field
        buf = np.zeros(64 * 64 * 4, np.uint8)
        assert lib.smol_cpu_scale_rows(ctypes.byref(bad), buf.ctypes.data, 256, buf.ctypes.data, 128, 0, 1) < 0, field
    # a box whose last span would end outside the image (64 -> 64 with a step of 2 pixels reads up to pixel 126)
    box = H.plan_job(0, 640, 64, 0, 64, 64, 0)
    assert box.filter_h == sm.SMOL_HIP_FILTER_BOX
    for field, value in (("width_in", 64), ("width_out", 641)):
        bad = sm.SmolHipJob.from_buffer_copy(bytes(box))
        setattr(bad, field, value)
        plan = ctypes.c_void_p()
        assert lib.smol_cpu_plan_create(ctypes.byref(bad), ctypes.byref(plan)) < 0, field


def test_cpu_backend_cuts_a_large_call_into_strips_over_the_host_cores():
    """Round 5: a call of enough rows runs as strips of output rows on a small persistent pool (SMOL_CPU_THREADS, default the
    online cores; the reference's own harness cuts an image the same way: test.c:838-883). Byte-equal to the oracle for the
    SIMD class and for generic classes with a vertical box and a bilinear halo; callers that come while the pool is busy run
    on their own thread."""
    import threading
    import time
    cases = [(0, 2048, 2048, 0, 512, 512, 0),          # the SIMD class (exact 4:1)
             (1, 1500, 1200, 2, 1000, 777, 0),         # SIMD class, bilinear halo between strips
             (8, 1600, 1400, 5, 150, 120, 0),          # box x box, 24-bit in, unassociated out
             (4, 900, 700, 6, 1300, 1000, 1)]          # premul16 + sRGB magnification
    for spec in cases:
        t_in, w_in, h_in, t_out, w_out, h_out, srgb = spec
        src = H.make_input(t_in, w_in, h_in, seed=3, valid=True, pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb), want), spec
        first, n = h_out // 5, h_out // 2 + 3
        rb = w_out * H.bpp(t_out)
        assert np.array_equal(cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, first=first, n=n), want[first * rb:(first + n) * rb]), spec
    # several callers at once: one gets the pool, the others their own thread
    t_in, w_in, h_in, t_out, w_out, h_out, srgb = cases[1]
    src = H.make_input(t_in, w_in, h_in, seed=3, valid=True, pad=16)
    want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    bad = []

    def worker():
        for _ in range(3):
            if not np.array_equal(cpu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb), want):
                bad.append(1)

    ts = [threading.Thread(target=worker) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad
