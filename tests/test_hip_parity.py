"""Parity tests proper: the HIP path (through the C-ABI) against the oracle on the
same seeded inputs, and against the committed golden vectors. Bit-exact: the path
is integer fixed-point, so the tolerance is zero bytes."""
import ctypes
import random
import threading

import numpy as np
import pytest

import cases as C
import helpers as H
import smolscale_amd as sm
from test_oracle_vs_reference import random_case

pytestmark = pytest.mark.gpu


def assert_same(got, want, h, row_bytes, stride, ctx):
    if np.array_equal(got, want):
        return
    G, W = H.rows_view(got, h, row_bytes, stride), H.rows_view(want, h, row_bytes, stride)
    nz = np.argwhere(G != W)
    if len(nz) == 0:
        raise AssertionError("row padding bytes were modified: %r" % (ctx,))
    raise AssertionError("%r: %d differing bytes, first at row %d byte %d: got %d want %d"
                         % (ctx, len(nz), nz[0][0], nz[0][1], G[tuple(nz[0])], W[tuple(nz[0])]))


def test_gpu_present_and_native_library_loaded():
    assert sm.lib().smol_hip_device_count() >= 1
    assert H.os.path.exists(sm.LIB_PATH)


def test_golden_matrix_every_type_pair_and_filter_class():
    """2800 reference-captured vectors: 10x10 type pairs x srgb x 14 geometry classes."""
    n = 0
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid), want in C.matrix_cases():
        src = H.make_input(t_in, w_in, h_in, seed=seed, valid=valid, pad=16)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        n += 1
    assert n == 2800


def test_random_sweep_vs_oracle_host_pointers():
    rng = random.Random(4242)
    for it in range(250):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 600_000)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        assert_same(got, want, n, w_out * H.bpp(t_out), sout,
                    (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n))


def test_random_sweep_vs_oracle_device_pointers_unaligned():
    """Caller-owned device buffers with odd strides and misaligned bases."""
    rng = random.Random(777)
    for it in range(120):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout = random_case(rng, 300_000)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), stride=sin, pad=16)
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n,
                                 src_misalign=rng.choice([0, 1, 2, 3, 5, 8]), dst_misalign=rng.choice([0, 1, 2, 3, 4]))
        assert_same(got, want, n, w_out * H.bpp(t_out), sout,
                    (t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, sout, first, n))


def test_pow2_fast_path_vs_oracle():
    rng = random.Random(31)
    for it in range(60):
        hx, hy = rng.randrange(3), rng.randrange(3)
        w_out = rng.choice([2, 4, 6, 64, 66, 130, 256, 300, 514, 1024])
        h_out = rng.randint(1, 40)
        w_in, h_in = w_out << (hx + 1), h_out << (hy + 1)
        t_in, t_out = rng.randrange(4), rng.randrange(4)
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out)
        assert np.array_equal(got, want), (hx, hy, w_out, h_out, t_in, t_out)


def test_pow2_streaming_kernels_on_cropped_and_odd_stride_sources():
    """Exact 2:1 / 4:1 / 8:1 from a sub-rectangle of a larger image (pixels_in + 4 * x, the parent's stride:
    smolscale.h:56-60) and from rows whose stride is not a multiple of 16: the streaming kernels take any
    4-byte aligned source (smolscale-generic.c:1623-1634 tolerates any alignment)."""
    rng = random.Random(3131)
    for it in range(80):
        h = rng.randrange(3)
        w_out = rng.choice([4, 8, 64, 72, 136, 256, 300, 1024])
        h_out = rng.randint(1, 40)
        w_in, h_in = w_out << (h + 1), h_out << (h + 1)
        t_in, t_out = rng.randrange(8), rng.randrange(10)
        x0 = rng.choice([1, 2, 3, 5, 7])                            # crop origin in the parent, pixels
        parent_w = w_in + x0 + rng.choice([0, 1, 2, 3, 4, 11])
        sin = parent_w * 4
        parent = H.make_input(t_in, parent_w, h_in, seed=it, valid=bool(it & 1), pad=32)
        src = parent[x0 * 4:]
        first = rng.randint(0, h_out - 1)
        n = rng.randint(1, h_out - first)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, None, first, n)
        # the device copy keeps the parent's 16-byte phase: base = aligned + 4 * x0
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, None, first, n,
                                 src_misalign=(4 * x0) % 16)
        assert np.array_equal(got, want), (h, w_out, h_out, t_in, t_out, x0, parent_w, first, n)


def test_one_tile_wide_and_tall_images_under_the_xcd_aware_order():
    """Launches of 64 and more workgroups run in the XCD-aware order, whose tile index is decoded with host-made
    reciprocals: images ONE tile wide (gx = 1: the reciprocal of 1 does not fit 32 bits) and a few tiles wide,
    thousands of row tiles tall -- the shape of verify.c's vertical ordering test (1 x 16384 -> 1 x 16383,
    verify.c:163-166) -- through k_tile, k_pow2 and k_pow2x, plus batches of many small images (gx * gy = 1)."""
    cases = [
        (0, 1, 16384, 8, 1, 16383),          # verify.c: V rgba -> rgb
        (8, 1, 16384, 0, 1, 16383),
        (0, 3, 9000, 1, 3, 4100),            # bilinear V, copy H
        (0, 40, 20000, 4, 17, 7000),         # one column tile, bilinear both
        (0, 70, 12000, 0, 65, 5000),         # two column tiles
        (0, 128, 32768, 0, 32, 8192),        # exact 4:1, one column tile: k_pow2
        (4, 128, 16384, 9, 64, 8192),        # exact 2:1, mixed types: k_pow2x
        (0, 600, 9000, 0, 40, 600),          # box both axes (ratio 15): two passes
        (0, 500, 4500, 0, 50, 450),          # box both axes (ratio 9-10): k_tile
    ]
    for (t_in, w_in, h_in, t_out, w_out, h_out) in cases:
        src = H.make_input(t_in, w_in, h_in, seed=w_in, valid=True, pad=32)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out)
        assert np.array_equal(got, want), ("host", t_in, w_in, h_in, t_out, w_out, h_out)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, src_misalign=4)
        assert np.array_equal(got, want), ("device+4", t_in, w_in, h_in, t_out, w_out, h_out)
    # many one-tile images in one launch
    rng = random.Random(9)
    specs, images, bufs, wants = [], [], [], []
    for i in range(200):
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        w_in, h_in = rng.randint(20, 60), rng.randint(20, 60)
        w_out, h_out = rng.randint(5, 19), rng.randint(5, 15)
        src = H.make_input(t_in, w_in, h_in, seed=i, valid=True, pad=16)
        dsrc, ddst = sm.DeviceBuffer(src.size + 16), sm.DeviceBuffer(w_out * h_out * H.bpp(t_out) + 16)
        dsrc.upload(src)
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
        im.src, im.dst = dsrc.ptr, ddst.ptr
        im.rowstride_in, im.rowstride_out = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
        im.first_row, im.n_rows = 0, h_out
        images.append(im)
        bufs.append((dsrc, ddst))
        specs.append((t_in, w_in, h_in, t_out, w_out, h_out))
        wants.append(H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out))
    b = sm.HipBatch(images)
    b.launch()
    sm.check(sm.lib().smol_hip_device_synchronize(0), "sync")
    for spec, (dsrc, ddst), want in zip(specs, bufs, wants):
        assert np.array_equal(ddst.download(want.size), want), spec
    b.destroy()
    for a, d in bufs:
        a.free()
        d.free()


def test_24bit_output_rows_at_every_byte_alignment():
    """24-bit output whose rows start on any byte (every width that is not a multiple of 4, odd strides, a destination
    1-3 bytes past a dword): full-width tiles leave as aligned dwords shifted inside the wave (tile_store_cols), the
    run's first and last bytes singly. Through k_tile (1 and 4 columns per lane), k_up, k_hspan; bytes between the
    rows and around the image must stay untouched."""
    rng = random.Random(2424)
    shapes = [
        (1.3, 1.3), (1.9, 1.6), (2.5, 2.5), (3.7, 2.2), (6.0, 5.0), (9.5, 9.5),      # k_tile, bilinear and box
        (0.4, 0.5), (0.8, 0.9), (0.27, 0.3),                                          # magnification: k_tile / k_up
        (2.0, 1.0), (4.0, 1.0),                                                       # k_hspan (vertical copy)
    ]
    for it in range(66):
        rx, ry = shapes[it % len(shapes)]
        w_out = rng.choice([256, 257, 258, 259, 300, 513, 1025])
        h_out = rng.randint(3, 40)
        w_in, h_in = max(2, int(round(w_out * rx))), max(2, int(round(h_out * ry)))
        if ry == 1.0:
            h_in = h_out
        t_in, t_out = rng.choice([0, 2, 4, 8, 9]), rng.choice([8, 9])
        sout = w_out * 3 + rng.choice([0, 1, 2, 3, 7])
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=True, pad=32)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, None, sout, fill=0xA5)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, None, sout, fill=0xA5,
                                 dst_misalign=it % 4)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sout, it % 4)


def test_pow2x_every_type_pair_and_storage_class():
    """k_pow2x (exact 2:1/4:1/8:1, any pixel types, 64bpp and 128bpp classes) through
    the host-pointer API and from caller-owned device buffers."""
    rng = random.Random(313)
    for it in range(200):
        h = rng.randrange(3)
        w_out = rng.choice([4, 8, 16, 24, 64, 72, 136, 256, 300, 1024])
        h_out = rng.randint(1, 70)
        w_in, h_in = w_out << (h + 1), h_out << (h + 1)
        t_in, t_out = (it // 10) % 10, it % 10
        srgb = (it // 100) & 1
        src = H.make_input(t_in, w_in, h_in, seed=it, valid=bool(it & 1), pad=32)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), ("host", h, w_out, h_out, t_in, t_out, srgb)
        if it % 4 == 0:
            first = rng.randint(0, h_out - 1)
            n = rng.randint(1, h_out - first)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, None, None, first, n)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, None, None, first, n)
            assert np.array_equal(got, want), ("device", h, w_out, h_out, t_in, t_out, srgb, first, n)


def test_up_kernel_magnification_shapes():
    """k_up: magnification on both axes; shapes that make tiles of 8..64 rows, spans
    from 2 to 258 source pixels, partial edge tiles, all output kinds, misaligned
    device buffers and odd strides."""
    shapes = [
        (4, 480, 270, 0, 1920, 1080), (0, 1000, 700, 1, 1010, 707), (8, 2, 2, 2, 900, 37),
        (0, 300, 200, 8, 2700, 1800), (9, 257, 65, 4, 1300, 300), (5, 64, 64, 5, 513, 511),
        (1, 1279, 719, 3, 1280, 720), (8, 100, 100, 9, 777, 333), (6, 33, 17, 0, 259, 130),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for (sin_pad, mis) in ((0, 0), (3, 1), (16, 4)):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host", sin_pad)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "device", sin_pad, mis)


def test_down_kernel_premul16_minification_shapes():
    """k_lind: unassociated -> unassociated minification at arbitrary ratios (0-2 halvings
    per axis, 1 and 4 columns per lane), host pointers and misaligned device buffers."""
    shapes = [
        (4, 4096, 300, 4, 2731, 200), (5, 1000, 700, 6, 333, 100), (7, 2000, 64, 4, 260, 9),
        (4, 515, 129, 5, 256, 65), (6, 4000, 33, 7, 1600, 11), (4, 799, 601, 4, 100, 99),
        (5, 64, 4000, 5, 9, 501), (4, 3, 3, 4, 2, 2),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for (sin_pad, mis) in ((0, 0), (4, 4), (16, 1)):
            sin = w_in * 4 + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=False, stride=sin, pad=0)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host", sin_pad)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "device", sin_pad, mis)


def test_up_kernel_srgb_linear_flavours():
    """k_up flavours 2 and 3 (with_srgb = 1 magnification), host and misaligned device buffers."""
    shapes = [
        (0, 480, 270, 0, 1920, 1080), (8, 300, 200, 8, 1300, 900), (4, 257, 65, 4, 1300, 300),
        (9, 64, 64, 5, 513, 511), (5, 100, 100, 2, 777, 333), (4, 2, 2, 6, 300, 37), (0, 1279, 719, 8, 1280, 720),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for (sin_pad, mis) in ((0, 0), (3, 1)):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host", sin_pad)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "device", sin_pad, mis)


def test_down_kernel_srgb_linear_flavours():
    """k_lind (with_srgb = 1): premultiplied, unassociated
    and 24-bit types, odd strides, misaligned device buffers."""
    shapes = [
        (0, 4096, 300, 0, 2731, 200), (8, 4000, 90, 8, 1600, 36), (4, 1000, 700, 6, 333, 100),
        (9, 2000, 64, 2, 260, 9), (1, 515, 129, 5, 256, 65), (4, 799, 601, 4, 100, 99), (8, 3, 3, 0, 2, 2),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for (sin_pad, mis) in ((0, 0), (3, 1)):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host", sin_pad)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "device", sin_pad, mis)


def test_axes_scaled_independently_in_the_128bpp_class():
    """Jobs whose axes differ (smol_jobdev.h: down_setup, twopass_setup): one minified and one magnified, a COPY axis, an
    axis of one pixel -> k_lind with both axes in one launch; bilinear on one axis and box on the other -> k_boxh / k_linh
    + k_linv / k_boxv through the intermediate image. Gamma-correct types of every kind and premul16 with and without
    sRGB, tiles wider and taller than one wave's, odd strides, row strips, misaligned device buffers."""
    import smolscale_amd as sm
    shapes = [
        # (t_in, w_in, h_in, t_out, w_out, h_out, srgb, kernel)
        (0, 1500, 300, 0, 1000, 700, 1, "k_lind"),          # H down, V up
        (8, 300, 1400, 8, 1100, 500, 1, "k_lind"),          # H up, V down, 24-bit both sides
        (4, 1500, 256, 4, 1000, 256, 1, "k_lind"),          # V COPY, premul16 + sRGB
        (4, 700, 900, 6, 700, 400, 0, "k_lind"),            # H COPY, premul16
        (5, 1200, 1, 7, 500, 40, 0, "k_lind"),              # V ONE
        (2, 1, 600, 3, 300, 250, 1, "k_lind"),              # H ONE
        (4, 2100, 130, 4, 300, 517, 0, "k_lind"),           # 2 halvings down, x4 up: tall tiles
        (0, 4096, 300, 0, 300, 200, 1, "k_boxh_batch + k_linv"),
        (4, 300, 4000, 4, 900, 200, 0, "k_linh_batch (vertical box sums in registers) + k_boxv"),
        (6, 333, 2500, 5, 500, 200, 1, "k_linh_batch (vertical box sums in registers) + k_boxv"),   # premul16 + sRGB: 19-bit channels, never narrowed
        (9, 3000, 100, 1, 200, 333, 1, "k_boxh_batch + k_linv"),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb, kernel) in enumerate(shapes):
        name = sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb))
        assert name.startswith(kernel), (name, kernel, shapes[k])
        for (sin_pad, mis, first, n) in ((0, 0, 0, h_out), (3 if H.bpp(t_in) == 3 else 4, 1, h_out // 3, h_out // 2), (16, 4, h_out - 1, 1)):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=bool(k & 1), stride=sin, pad=0)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, None, first, n)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, None, first, n)
            assert np.array_equal(got, want), (shapes[k], "host", sin_pad, first, n)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, None, first, n,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (shapes[k], "device", sin_pad, mis, first, n)


def test_gamma_correct_kernels_on_opaque_images_and_on_a_translucent_block():
    """with_srgb = 1 on what gamma-correct scaling is used for -- photographs with an alpha channel. Every wave of an
    all-opaque image takes the fused one-look-up conversions (WU_LIN_O / WP_LIN_O, chosen per wave by a ballot: per row
    in k_lind, per batch in k_upg / k_pow2x, per group with hysteresis in k_boxh); with one translucent block in the
    image the waves over its border decide batch by batch. 4:1 (k_pow2x), x3 (k_upg), ratios 1.5 and 2.5 (k_lind), box
    ratios 9 and 20 (k_boxh + k_boxv); premultiplied, unassociated and 24-bit destinations; host and device entries."""
    shapes = [(1024, 256, 256, 64), (200, 60, 600, 180), (900, 300, 600, 200), (1000, 250, 400, 100),
              (1800, 180, 200, 20), (2000, 400, 100, 20)]
    pairs = [(0, 0), (2, 5), (4, 3), (5, 4), (7, 8), (1, 9), (6, 6)]
    k = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for (t_in, t_out) in pairs[k % 2::2] if (w_in * h_in > 500_000) else pairs:
            for block in (False, True):
                k += 1
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, pad=0)
                px = src[:w_in * h_in * 4].reshape(h_in, w_in, 4)
                apos = 3 if (t_in & 3) < 2 else 0
                keep = px[:, :, apos].copy()
                px[:, :, apos] = 255
                if block:
                    y0, x0 = h_in // 3, w_in // 4
                    px[y0:y0 + h_in // 4 + 1, x0:x0 + w_in // 3 + 1, apos] = keep[y0:y0 + h_in // 4 + 1, x0:x0 + w_in // 3 + 1]
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1)
                assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, block, "host")
                got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 1)
                assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, block, "device")


def test_two_pass_box_thumbnails_single_and_batched():
    """k_boxh + k_boxv (strong minification; every 128bpp-class box): host API, device API
    (temporary intermediate image), one batch with its own scratch, and the same batch as a hipGraph."""
    shapes = [
        (8, 4000, 300, 8, 200, 15, 0), (8, 4000, 300, 8, 200, 15, 1), (4, 2000, 900, 4, 64, 30, 0),
        (0, 3000, 2000, 1, 100, 5, 1), (5, 999, 777, 9, 1, 1, 1), (9, 1300, 500, 2, 65, 9, 0),
        # boxes wider than 255 pixels (128bpp class): 8 lanes per column, and the whole wave on one column from a ratio of
        # 400 (its own kernel: the batch launches the two flavours separately)
        (0, 9000, 40, 0, 30, 4, 0), (2, 20000, 240, 3, 9, 3, 0), (8, 30000, 9, 0, 40, 2, 1), (4, 65535, 12, 4, 64, 3, 0),
    ]
    lib = sm.lib()
    images, bufs, wants = [], [], []
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in enumerate(shapes):
        sin = w_in * H.bpp(t_in) + (3 if k & 1 else 0)
        src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin)
        assert np.array_equal(H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin), want), ("host", k)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, src_misalign=k & 3, dst_misalign=k & 1)
        assert np.array_equal(got, want), ("device", k)
        dsrc = sm.DeviceBuffer(src.size + 16)
        dsrc.upload(src)
        ddst = sm.DeviceBuffer(want.size + 16)
        ddst.fill(0)
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        im.src, im.dst = dsrc.ptr, ddst.ptr
        im.rowstride_in, im.rowstride_out = sin, w_out * H.bpp(t_out)
        im.first_row, im.n_rows = 0, h_out
        images.append(im)
        bufs.append((dsrc, ddst))
        wants.append(want)
    batch = sm.HipBatch(images)
    for use_graph in (False, True):
        batch.use_graph(use_graph)
        batch.launch()
        sm.check(lib.smol_hip_device_synchronize(0))
        for k, (want, (dsrc, ddst)) in enumerate(zip(wants, bufs)):
            assert np.array_equal(ddst.download(want.size), want), ("batch", use_graph, k)
            ddst.fill(0)
    batch.destroy()
    for dsrc, ddst in bufs:
        dsrc.free()
        ddst.free()


def test_two_pass_batch_mixes_pixel_types_and_gamma_at_one_geometry():
    """the two k_lin* passes take their flavour (11-bit, 24-bit rows, 19-bit, premul16) from the
    launch group's representative, so a batch that mixes premultiplied, unassociated (premul16) and gamma-correct images of ONE
    geometry must not put them into one group. Bilinear H x tall box V (vertical sums in pass 1's registers) and box H x
    bilinear V, every image against the oracle."""
    lib = sm.lib()
    for (w_in, h_in, w_out, h_out) in ((300, 3000, 200, 10), (3000, 300, 10, 200)):
        flavours = [(0, 0, 0), (5, 5, 0), (5, 5, 1), (0, 0, 1), (8, 8, 1), (8, 0, 0), (6, 7, 1), (1, 0, 0)]
        images, bufs, wants = [], [], []
        for k, (t_in, t_out, srgb) in enumerate(flavours):
            src = H.make_input(t_in, w_in, h_in, seed=70 + k, valid=True, pad=0)
            wants.append(H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb))
            dsrc = sm.DeviceBuffer(src.size + 16)
            dsrc.upload(src)
            ddst = sm.DeviceBuffer(wants[-1].size + 16)
            ddst.fill(0)
            im = sm.SmolHipImage()
            im.job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
            im.src, im.dst = dsrc.ptr, ddst.ptr
            im.rowstride_in, im.rowstride_out = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
            im.first_row, im.n_rows = 0, h_out
            images.append(im)
            bufs.append((dsrc, ddst))
        batch = sm.HipBatch(images)
        batch.launch()
        sm.check(lib.smol_hip_device_synchronize(0))
        for k, (want, (dsrc, ddst)) in enumerate(zip(wants, bufs)):
            assert np.array_equal(ddst.download(want.size), want), (w_in, h_in, w_out, h_out, flavours[k])
        batch.destroy()
        for dsrc, ddst in bufs:
            dsrc.free()
            ddst.free()


def test_host_strip_in_chunks_keeps_the_splits_of_its_tall_box():
    """bilinear H x tall box V keeps per-split partial sums in the lane's intermediate image, sized
    from the strip; the strip's chunks (fewer rows each) would pick more splits on their own. 96 x 32768 -> 400 x 512 from
    pageable memory is cut into chunks on both host roads; gamma-correct and premul16, whole image and a strip."""
    for (t_in, t_out, srgb) in ((5, 5, 1), (4, 4, 0), (0, 0, 1)):
        w_in, h_in, w_out, h_out = 96, 32768, 400, 512
        src = H.make_input(t_in, w_in, h_in, seed=9, valid=True, pad=0)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, t_out, srgb)


def test_boxes_wider_than_the_lds_read_straight_from_memory():
    """Round 6: horizontal boxes of more than ~9 600 pixels per column (a 65 535-pixel row boxed into one pixel) left the
    backstop kernels: k_boxh reads them straight from memory (boxh_direct_body). Host API, device API with misaligned
    pointers, and all of them as one batch."""
    from test_emu_kernels import WIDE_SPAN_CASES
    lib = sm.lib()
    images, bufs, wants = [], [], []
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in enumerate(WIDE_SPAN_CASES):
        sin = w_in * H.bpp(t_in) + (k & 1)
        src = H.make_input(t_in, w_in, h_in, seed=k, valid=bool(k & 1), stride=sin, pad=0)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin)
        name = sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb), first_row=0, n_rows=h_out)
        assert "k_boxh" in name, (name, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin), want), ("host", k)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, src_misalign=k % 3, dst_misalign=k & 1)
        assert np.array_equal(got, want), ("device", k)
        dsrc = sm.DeviceBuffer(src.size + 16)
        dsrc.upload(src)
        ddst = sm.DeviceBuffer(want.size + 16)
        ddst.fill(0)
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        im.src, im.dst = dsrc.ptr, ddst.ptr
        im.rowstride_in, im.rowstride_out = sin, w_out * H.bpp(t_out)
        im.first_row, im.n_rows = 0, h_out
        images.append(im)
        bufs.append((dsrc, ddst))
        wants.append(want)
    batch = sm.HipBatch(images)
    batch.launch()
    sm.check(lib.smol_hip_device_synchronize(0))
    for k, (want, (dsrc, ddst)) in enumerate(zip(wants, bufs)):
        assert np.array_equal(ddst.download(want.size), want), ("batch", k)
        dsrc.free()
        ddst.free()
    batch.destroy()


def test_box_filter_single_pass_in_launches_of_150_mb_and_more():
    """64bpp class, box on both axes: one image or a few go through the two passes (k_boxh + k_boxv), a launch that reads
    150 MB or more at ratios up to 12 (300 MB up to 24; 24-bit sources up to 19) through the single-pass k_tile
    (smol_jobdev.h: box2_setup). Both roads on the same images, every output against the oracle."""
    import smolscale_amd as sm
    lib = sm.lib()
    for (t_in, t_out, w_in, h_in, w_out, h_out, n_few, n_many) in ((0, 1, 4096, 2600, 430, 273, 1, 4), (8, 8, 4000, 3000, 380, 285, 2, 5),
                                                                 (2, 0, 4096, 3072, 228, 171, 1, 7)):
        srcs = [H.make_input(t_in, w_in, h_in, seed=40 + i, valid=True, pad=0) for i in range(2)]
        wants = [H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0) for src in srcs]
        dsrcs = []
        for src in srcs:
            d = sm.DeviceBuffer(src.size + 16)
            d.upload(src)
            dsrcs.append(d)
        for n in (n_few, n_many):
            images, ddsts = [], []
            for i in range(n):
                ddst = sm.DeviceBuffer(wants[0].size + 16)
                ddst.fill(0)
                im = sm.SmolHipImage()
                im.job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                im.src, im.dst = dsrcs[i & 1].ptr, ddst.ptr
                im.rowstride_in, im.rowstride_out = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
                im.first_row, im.n_rows = 0, h_out
                images.append(im)
                ddsts.append(ddst)
            batch = sm.HipBatch(images)
            kname = batch.info()[1]
            assert ("k_tile" in kname) == (n == n_many), (kname, n, t_in, w_in)
            batch.launch()
            sm.check(lib.smol_hip_device_synchronize(0))
            for i, ddst in enumerate(ddsts):
                assert np.array_equal(ddst.download(wants[0].size), wants[i & 1]), (kname, n, i, t_in)
                ddst.free()
            batch.destroy()
        for d in dsrcs:
            d.free()


def test_workgroups_with_more_than_64k_of_lds():
    """Geometries whose workgroups need 70-130 KB of dynamic LDS (gfx950 has 160 KB per CU)."""
    cases = [
        (4, 2000, 100, 4, 2050, 110, 1),    # k_upg<32>, mild magnification
        (0, 4000, 600, 0, 64, 10, 0),       # k_boxh, 64bpp class, ratio 62
        (0, 4000, 600, 0, 64, 10, 1),       # k_boxh, 128bpp class, ratio 62
        (8, 4000, 100, 8, 3990, 99, 1),     # k_lind, mild minification
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in enumerate(cases):
        src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, pad=0)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def test_tile_kernel_wide_images_and_box_shapes():
    """Geometries that select every k_tile layout: 4 columns per lane (wide, few taps),
    one column per lane, box with 1 and with 8 lanes per column, box on one axis only."""
    shapes = [
        (4, 480, 270, 0, 1920, 1080), (0, 1000, 700, 1, 1000, 300), (8, 1, 37, 2, 900, 37),
        (0, 3000, 900, 0, 1700, 500), (9, 2000, 300, 0, 300, 100), (1, 2500, 1000, 5, 330, 330),
        (0, 4096, 1024, 0, 448, 112), (2, 4000, 120, 3, 20, 120), (8, 3000, 600, 0, 100, 7),
        (0, 640, 4000, 4, 320, 40), (5, 900, 901, 0, 100, 450), (0, 5000, 64, 8, 25, 64),
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for (sin_pad, mis) in ((0, 0), (3, 1)):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=32)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host", sin_pad)
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin,
                                     src_misalign=mis, dst_misalign=mis)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "device", sin_pad, mis)


def test_tile_kernel_vertical_modes_and_staging_variants():
    """One case per k_tile entry point added in round 2: the vertical mode compiled in (bilinear with one
    and four columns per lane at 0, 1 and 2 halvings; box with 1 and 8 lanes per column; copy), the
    run-time fallback (bilinear on one axis, box on the other), unassociated 32-bit sources premultiplied
    in LDS after the DMA, unassociated output through the LDS inverse table, tiles of 63 vertical samples
    (the one-register schedule) and the new tile heights (many tile rows, ragged last tile)."""
    shapes = [
        # (t_in, w_in, h_in, t_out, w_out, h_out)
        (0, 1500, 700, 0, 1000, 467),      # bilinear/bilinear 0H, one column per lane (ratio 1.5)
        (0, 1500, 700, 0, 1200, 600),      # ratio 1.25: 16 rows per tile
        (0, 1500, 701, 0, 800, 370),       # ratio 1.9: four columns per lane
        (0, 1800, 900, 0, 600, 301),       # 1H both axes
        (1, 2600, 1300, 2, 433, 216),      # 2H both axes, channel reorder
        (8, 2000, 1500, 8, 800, 600),      # 24-bit in and out, four columns per lane
        (9, 2000, 1500, 0, 1333, 1000),    # 24-bit in, 32-bit out, 0H
        (4, 2000, 1500, 0, 800, 600),      # unassociated in (premultiplied in LDS), 1H
        (5, 1500, 700, 1, 1000, 467),      # unassociated in, 0H, four columns per lane
        (0, 1800, 900, 4, 600, 301),       # unassociated out (LDS inverse table), dword stores
        (8, 1800, 900, 6, 600, 301),       # 24-bit in, unassociated out
        (0, 1500, 700, 8, 1000, 467),      # 24-bit out from 32-bit in
        (0, 2000, 2000, 0, 200, 190),      # box/box, one lane per column
        (0, 6000, 900, 0, 150, 30),        # box/box, eight lanes per column
        (0, 2000, 900, 0, 700, 30),        # bilinear H, box V: run-time vertical mode
        (0, 6000, 300, 0, 150, 120),       # box H, bilinear V
        (0, 1500, 64, 0, 1000, 200),       # H down, V up (32-row tiles of a magnified axis)
        (0, 1500, 700, 0, 1000, 700),      # vertical copy
    ]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out) in enumerate(shapes):
        for sin_pad in (0, 4, 5):
            sin = w_in * H.bpp(t_in) + sin_pad
            src = H.make_input(t_in, w_in, h_in, seed=40 + k, valid=False, stride=sin, pad=32)
            want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin)
            assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, sin_pad)
        # a strip that starts and ends inside tiles
        first, n = h_out // 3, h_out // 2
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, None, first, n)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, None, first, n)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "strip")


def test_strip_split_determinism():
    """Any partition of the output rows gives the one-shot bytes."""
    rng = random.Random(9)
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (0, 640, 480, 0, 320, 240, 0), (4, 200, 150, 0, 333, 401, 0), (0, 900, 700, 5, 90, 70, 0),
            (8, 513, 300, 2, 64, 300, 0), (6, 128, 128, 7, 32, 32, 1), (0, 700, 3000, 0, 2, 3, 0)):
        src = H.make_input(t_in, w_in, h_in, seed=3, valid=False, pad=16)
        full = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(full, H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb))
        stride = w_out * H.bpp(t_out)
        cuts = sorted(set([0, h_out] + [rng.randint(0, h_out) for _ in range(6)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, first=a, n=b - a)
            assert np.array_equal(part, full[a * stride:b * stride]), (t_in, t_out, a, b)


def test_baseline_configs_against_golden_digests():
    g = C.load_golden("baseline.npz")
    for name in ("C1", "C2", "C3", "C5", "BOX", "C4"):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, valid, stride_in = [int(v) for v in g[name + "_params"]]
        src = H.make_input(t_in, w_in, h_in, seed=0, valid=bool(valid), stride=stride_in, pad=16)
        out = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, stride_in=stride_in)
        bo = H.bpp(t_out)
        assert H.fnv1a(out, w_out * bo, h_out, w_out * bo) == int(g[name + "_fnv"][0]), name
        cw, ch = min(64, w_out), min(64, h_out)
        x0, y0 = (w_out - cw) // 2, (h_out - ch) // 2
        rows = H.rows_view(out, h_out, w_out * bo, w_out * bo)
        assert np.array_equal(rows[y0:y0 + ch, x0 * bo:(x0 + cw) * bo], g[name + "_crop"]), name
        del src, out


def test_north_star_config_full_frame_vs_oracle():
    """4096^2 -> 1024^2 RGBA8 premultiplied, every byte, plus linearity-free properties."""
    src = H.make_input(0, 4096, 4096, seed=5, valid=True, pad=16)
    want = H.oracle_scale(src, 0, 4096, 4096, 0, 1024, 1024)
    got = H.hip_scale(src, 0, 4096, 4096, 0, 1024, 1024)
    assert np.array_equal(got, want)
    # solid colour stays that colour (test.c:1128-1192 invariant), idempotent channel swap
    solid = np.tile(np.array([4, 5, 6, 7], np.uint8), 4096 * 4096)
    out = H.hip_scale(solid, 2, 4096, 4096, 2, 1024, 1024)
    assert np.array_equal(out.reshape(-1, 4), np.tile(np.array([4, 5, 6, 7], np.uint8), (1024 * 1024, 1)))


def test_solid_colour_invariant_all_filter_classes():
    """test.c:1194-1298 restated on a pruned grid: premultiplied ARGB solids
    (i,i+1,i+2,i+3). HIP must equal the oracle everywhere; where the reference
    keeps the colour exactly (all bilinear/copy/one classes), so must HIP."""
    sizes = [(1, 1, 5, 5), (7, 7, 7, 7), (300, 200, 640, 480), (640, 480, 320, 240), (1000, 900, 300, 250),
             (1000, 1000, 130, 140), (4096, 64, 448, 7), (2000, 600, 3, 2), (65535, 2, 1, 1), (2, 65535, 1, 7)]
    for (w_in, h_in, w_out, h_out) in sizes:
        job = H.plan_job(H.ARGB8_P, w_in, h_in, H.ARGB8_P, w_out, h_out)
        has_box = H.FILTER_BOX in (job.filter_h, job.filter_v)
        for i in (0, 60, 128, 252):
            px = np.array([i + 3, i, i + 1, i + 2], np.uint8)      # A first; colour <= alpha
            src = np.concatenate([np.tile(px, w_in * h_in), np.zeros(16, np.uint8)])
            out = H.hip_scale(src, H.ARGB8_P, w_in, h_in, H.ARGB8_P, w_out, h_out)
            want = H.oracle_scale(src, H.ARGB8_P, w_in, h_in, H.ARGB8_P, w_out, h_out)
            assert np.array_equal(out, want), (w_in, h_in, w_out, h_out, i)
            if not has_box:
                assert np.array_equal(out.reshape(-1, 4), np.tile(px, (w_out * h_out, 1))), (w_in, h_in, w_out, h_out, i)


def test_batched_images_mixed_types_single_call():
    """Config 5 shape: many 256^2 -> 64^2 thumbnails with mixed pixel types in one batch."""
    rng = random.Random(55)
    n_img = 48
    lib = sm.lib()
    specs, srcs, images, dbufs = [], [], [], []
    for i in range(n_img):
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        src = H.make_input(t_in, 256, 256, seed=i, valid=True, pad=16)
        dsrc = sm.DeviceBuffer(src.size)
        dsrc.upload(src)
        ddst = sm.DeviceBuffer(64 * 64 * H.bpp(t_out))
        ddst.fill(0)
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, 256, 256, t_out, 64, 64, 0)
        im.src, im.dst = dsrc.ptr, ddst.ptr
        im.rowstride_in, im.rowstride_out = 256 * H.bpp(t_in), 64 * H.bpp(t_out)
        im.first_row, im.n_rows = 0, 64
        specs.append((t_in, t_out))
        srcs.append(src)
        images.append(im)
        dbufs.append((dsrc, ddst))
    batch = sm.HipBatch(images)
    n_launches, name = batch.info()
    assert 1 <= n_launches <= 16 and name      # one launch per kernel entry point in the mix
    batch.launch()
    sm.check(lib.smol_hip_device_synchronize(0))
    wants = [H.oracle_scale(src, t_in, 256, 256, t_out, 64, 64) for (t_in, t_out), src in zip(specs, srcs)]
    for want, (dsrc, ddst), spec in zip(wants, dbufs, specs):
        assert np.array_equal(ddst.download(), want), spec
        ddst.fill(0)
    # the same batch as a hipGraph of per-image kernel nodes
    batch.use_graph(True)
    batch.launch()
    sm.check(lib.smol_hip_device_synchronize(0))
    for want, (dsrc, ddst), spec in zip(wants, dbufs, specs):
        assert np.array_equal(ddst.download(), want), spec
        dsrc.free()
        ddst.free()
    batch.destroy()


def test_ten_thousand_mixed_thumbnails_every_output_checked():
    """BASELINE config 5 at its full count: 10 000 thumbnails 256^2 -> 64^2, pixel types drawn from the whole
    10 x 10 table, ONE batch -- and every one of the 10 000 outputs compared with the oracle (each image has
    its own destination; the sources come from a pool of 8 per input type so that the oracle side is 800
    scalings, not 10 000)."""
    rng = random.Random(2025)
    n_img, pool = 10_000, 8
    lib = sm.lib()
    srcs = {(t, k): H.make_input(t, 256, 256, seed=100 * t + k, valid=bool(k & 1), pad=0)
            for t in range(10) for k in range(pool)}
    dsrc = {}
    for key, a in srcs.items():
        d = sm.DeviceBuffer(a.size)
        d.upload(a)
        dsrc[key] = d
    picks = [(rng.randrange(10), rng.randrange(pool), rng.randrange(10)) for _ in range(n_img)]
    out_off, total = [], 0
    for (t_in, k, t_out) in picks:
        out_off.append(total)
        total += (64 * 64 * H.bpp(t_out) + 255) & ~255
    ddst = sm.DeviceBuffer(total)
    ddst.fill(0x5A)
    images = []
    for (t_in, k, t_out), off in zip(picks, out_off):
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, 256, 256, t_out, 64, 64, 0)
        im.src, im.dst = dsrc[(t_in, k)].ptr, ddst.ptr + off
        im.rowstride_in, im.rowstride_out = 256 * H.bpp(t_in), 64 * H.bpp(t_out)
        im.first_row, im.n_rows = 0, 64
        images.append(im)
    batch = sm.HipBatch(images)
    batch.launch()
    sm.check(lib.smol_hip_device_synchronize(0))
    got = ddst.download()
    want = {}
    bad = []
    for i, ((t_in, k, t_out), off) in enumerate(zip(picks, out_off)):
        key = (t_in, k, t_out)
        if key not in want:
            want[key] = H.oracle_scale(srcs[(t_in, k)], t_in, 256, 256, t_out, 64, 64)
        w = want[key]
        if not np.array_equal(got[off:off + w.size], w):
            bad.append((i, key))
        # the padding between destinations stays untouched
        pad_end = off + ((w.size + 255) & ~255)
        if not (got[off + w.size:pad_end] == 0x5A).all():
            bad.append((i, key, "padding"))
    assert not bad, bad[:10]
    assert len(want) > 700
    # the same 10 000 images as BASELINE words config 5: a hipGraph of one kernel node per image
    # (smol_hip_batch_use_graph), every output compared again
    ddst.fill(0x5A)
    batch.use_graph(True)
    batch.launch()
    sm.check(lib.smol_hip_device_synchronize(0))
    got = ddst.download()
    bad = [(i, key) for i, ((t_in, k, t_out), off) in enumerate(zip(picks, out_off))
           for key in [(t_in, k, t_out)] if not np.array_equal(got[off:off + want[key].size], want[key])]
    assert not bad, ("graph", bad[:10])
    batch.destroy()
    for d in dsrc.values():
        d.free()
    ddst.free()


def test_post_row_callback_runs_per_row_in_order():
    seen = []

    def cb(row, width, user):
        seen.append(width)
        row[0] = 0x11223344

    src = H.make_input(0, 64, 48, seed=1, pad=16)
    dst = np.zeros(32 * 24 * 4, np.uint8)
    ctx = sm.SmolScaleCtx(src, 0, 64, 48, 256, dst, 0, 32, 24, 128, 0, post_row_func=cb)
    ctx.batch(0, 24)
    ctx.destroy()
    want = H.oracle_scale(src, 0, 64, 48, 0, 32, 24).reshape(24, 128).copy()
    want[:, 0:4] = np.frombuffer(np.uint32(0x11223344).tobytes(), np.uint8)
    assert seen == [32] * 24
    assert np.array_equal(dst.reshape(24, 128), want)


def test_smol_scale_simple_and_batch_write_in_place():
    src = H.make_input(1, 300, 200, seed=2, pad=16)
    dst = np.full(100 * 77 * 4, 0x5a, np.uint8)
    sm.smol_scale_simple(src, 1, 300, 200, 1200, dst, 3, 100, 77, 400, 0)
    assert np.array_equal(dst, H.oracle_scale(src, 1, 300, 200, 3, 100, 77))


def test_concurrent_batches_on_one_context():
    """smolscale.h:71-73: concurrent smol_scale_batch() on disjoint rows of one ctx."""
    src = H.make_input(0, 1600, 1200, seed=9, pad=16)
    dst = np.zeros(400 * 300 * 4, np.uint8)
    ctx = sm.SmolScaleCtx(src, 0, 1600, 1200, 6400, dst, 0, 400, 300, 1600, 0)
    n_threads = 8
    rows = (300 + n_threads - 1) // n_threads          # test.c:870 strip rule
    threads = [threading.Thread(target=ctx.batch, args=(i * rows, min(rows, 300 - i * rows)))
               for i in range(n_threads) if i * rows < 300]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ctx.destroy()
    assert np.array_equal(dst, H.oracle_scale(src, 0, 1600, 1200, 0, 400, 300))


def test_rows_sharded_over_several_device_entries(monkeypatch):
    """SMOL_HIP_DEVICES splits every batch into one strip per listed device, run
    concurrently; listing device 0 three times exercises the sharding on one GPU."""
    monkeypatch.setenv("SMOL_HIP_DEVICES", "0,0,0")
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (0, 1024, 1024, 0, 256, 256, 0), (4, 300, 200, 1, 777, 501, 0), (0, 2000, 1500, 0, 100, 77, 0),
            (6, 128, 128, 7, 32, 32, 1), (8, 513, 2, 2, 64, 2, 0)):
        src = H.make_input(t_in, w_in, h_in, seed=12, valid=True, pad=16)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)), (t_in, w_in, t_out, w_out)
        part = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, first=h_out // 3, n=h_out - h_out // 3)
        stride = w_out * H.bpp(t_out)
        assert np.array_equal(part, got[(h_out // 3) * stride:])


def test_row_shards_from_resident_slabs_with_virtual_row0():
    """What bench.py --gpus N and a multi-GPU scatter do: a rank holds only the slab of
    input rows its strip needs and passes the address of (virtual) input row 0."""
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (0, 4096, 1024, 0, 1024, 256, 0), (4, 640, 360, 0, 1920, 1080, 0), (8, 3000, 900, 1, 370, 111, 0),
            (0, 4096, 2048, 0, 448, 224, 0), (5, 512, 512, 6, 128, 128, 0)):
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        s_in, s_out = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
        full = H.make_input(t_in, w_in, h_in, seed=21, valid=True, pad=16)
        want = H.oracle_scale(full, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        world = 3
        images, bufs = [], []
        for r in range(world):
            first, n, r0, r1 = sm.shard(job, r, world)
            slab = full[r0 * s_in:(r1 + 1) * s_in]
            dsrc = sm.DeviceBuffer(slab.size + 64)
            dsrc.upload(slab)
            ddst = sm.DeviceBuffer(n * s_out + 64)
            im = sm.SmolHipImage()
            im.job = job
            im.src = dsrc.ptr - r0 * s_in
            im.dst = ddst.ptr
            im.rowstride_in, im.rowstride_out = s_in, s_out
            im.first_row, im.n_rows = first, n
            images.append(im)
            bufs.append((dsrc, ddst, first, n))
        batch = sm.HipBatch(images)
        batch.launch()
        sm.check(sm.lib().smol_hip_device_synchronize(0))
        for (dsrc, ddst, first, n) in bufs:
            got = ddst.download(n * s_out)
            assert np.array_equal(got, want[first * s_out:(first + n) * s_out]), (t_in, w_in, t_out, w_out, first, n)
            dsrc.free()
            ddst.free()
        batch.destroy()


def test_tall_vertical_box_with_few_output_pixels():
    """128bpp-class vertical boxes over a handful of output pixels run one wave per pixel."""
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (0, 2048, 2048, 0, 7, 7, 0), (4, 300, 9000, 5, 300, 20, 0), (0, 500, 4000, 0, 25, 50, 1),
            (8, 40, 65535, 0, 40, 3, 1), (6, 64, 3000, 7, 2, 9, 0)):
        src = H.make_input(t_in, w_in, h_in, seed=5, valid=True, pad=16)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)), \
            (t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def test_edge_sizes():
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (0, 1, 1, 0, 1, 1, 0), (8, 1, 1, 9, 65535, 1, 0), (0, 2, 1, 0, 1, 1, 1), (4, 65535, 1, 4, 1, 1, 1),
            (0, 65535, 3, 0, 65535, 3, 0), (9, 2, 2, 5, 3, 3, 1), (0, 3, 65535, 8, 3, 1, 0)):
        src = H.make_input(t_in, w_in, h_in, seed=w_in, valid=False, pad=16)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)), \
            (t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def test_tall_vertical_boxes_behind_a_bilinear_horizontal_axis():
    """VERDICT round 4, Next 5: a vertical box of hundreds or thousands of rows behind a mild horizontal axis used to write
    an intermediate image larger than the source (300 x 20000 -> 200 x 40 at 0.15 of the roofline, 64 x 60000 -> 100 x 4 at
    0.06). k_linh now keeps the vertical box sums in registers, one tile per (column tile, output row, SPLIT of the row's
    span), and k_boxv adds the splits' partial sums (JobDev::box2 == 3). Gamma-correct, premul16 and the class's compressed
    storage; strips; device pointers (strips of one image share the plan's buffer without overlapping)."""
    cases = [(0, 300, 20000, 0, 200, 40, 0),       # FLATBILBOX: compressed storage (box of 500 rows), H ratio 1.5
             (0, 64, 60000, 0, 100, 4, 0),         # FLATTALL: H magnified, V ratio 15000
             (5, 333, 9000, 6, 150, 7, 1),         # gamma-correct, unassociated in and out with sRGB: excluded (19-bit) -> backstop
             (1, 500, 6400, 2, 350, 50, 1),        # gamma-correct premultiplied, ratio 128
             (8, 257, 8000, 9, 400, 9, 1),         # 24-bit, H magnified
             (4, 640, 5000, 7, 640, 20, 0),        # premul16, H COPY
             (6, 1, 3000, 4, 33, 3, 0)]            # H ONE
    lib = sm.lib()
    lib.smol_hip_kernel_name.restype = ctypes.c_char_p
    seen = 0
    for spec in cases:
        t_in, w_in, h_in, t_out, w_out, h_out, srgb = spec
        src = H.make_input(t_in, w_in, h_in, seed=31, valid=True, pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        name = lib.smol_hip_kernel_name(ctypes.byref(job), ctypes.c_void_p(256), w_in * H.bpp(t_in), ctypes.c_void_p(256), w_out * H.bpp(t_out), 0, h_out).decode()
        seen += "box sums in registers" in name
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (spec, name)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        assert np.array_equal(got, want), (spec, name, "device")
        if h_out >= 4:
            first, n = h_out // 3, h_out // 2
            got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, first=first, n=n)
            rb = w_out * H.bpp(t_out)
            assert np.array_equal(got, want[first * rb:(first + n) * rb]), (spec, name, "strip")
    assert seen >= 5


def test_gamma_correct_24_bit_sources_three_channel_flavour():
    """Round 5: sources without an alpha channel take k_lind's three-channel flavour (lind3_body: SDWA + offset-field look-ups,
    v_mad_u32_u16 taps, alpha neither staged nor filtered) -- the wide tiles (two columns per lane), the one-column flavour
    behind one and two halvings, magnification, mixed axes; into every output type, 24-bit output through the quad dword
    stores at every row alignment (odd widths, odd strides, misaligned device buffers, tiles that end inside a quad)."""
    shapes = [(4000, 300, 1600, 120),      # ratio 2.5, 1 halving: wide tiles of 127 columns
              (3000, 210, 420, 30),        # ratio 7.1, 2 halvings: one-column flavour
              (1500, 200, 1000, 133),      # ratio 1.5, no halving
              (480, 100, 1601, 333),       # magnified x3.3: stores dominate (from x3.5 32-bit output is k_upg's)
              (1000, 64, 333, 200),        # H down, V up
              (257, 300, 1031, 75)]        # H up, V down
    lib = sm.lib()
    k = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for t_in in (8, 9):
            for t_out in (8, 9, 0, 3, 5):
                k += 1
                if (k % 3) and w_in * h_in > 600_000:
                    continue
                sin = w_in * 3 + (k % 4)
                sout = w_out * H.bpp(t_out) + (k % 3)
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
                assert sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 1)) == "k_lind_batch"
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin, sout)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin, sout)
                rb = w_out * H.bpp(t_out)
                assert np.array_equal(H.rows_view(got, h_out, rb, sout), H.rows_view(want, h_out, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "host")
                first, n = h_out // 3, max(1, h_out // 2)
                got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 1, sin, sout, first, n,
                                         src_misalign=k % 4, dst_misalign=(k // 2) % 4)
                assert np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want[first * sout:], n, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "device")


def test_magnification_into_24_bit_rows_through_k_up():
    """Round 5: k_up with 24-bit output -- the lane's four pixels leave as twelve bytes (an 8-byte and a 4-byte store at the
    pixels' own address) -- from x2.6, every 64bpp-class source type into RGB8 / BGR8: aligned and odd widths, padded and
    byte-misaligned rows, partial tiles at the right edge (k_tile's column stores), row ranges."""
    shapes = [(480, 270, 1920, 1080),      # x4: 7.5 tiles of 256 columns
              (333, 100, 1000, 301),       # x3, odd everything
              (100, 40, 261, 105),         # x2.61: one full tile and five columns
              (64, 20, 512, 161),          # x8
              (700, 20, 1900, 60)]         # x2.7, wide
    k = 0
    seen = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for t_in in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
            for t_out in (8, 9):
                k += 1
                if (k % 2) and w_in * h_in > 60_000:
                    continue
                job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                if job.wide:
                    continue
                seen += sm.kernel_name(job) == "k_up_batch"
                sin = w_in * H.bpp(t_in) + (k % 4)
                sout = w_out * 3 + (k % 3)
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                rb = w_out * 3
                assert np.array_equal(H.rows_view(got, h_out, rb, sout), H.rows_view(want, h_out, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "host")
                first, n = h_out // 3, max(1, h_out // 2)
                got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n,
                                         src_misalign=k % 3, dst_misalign=(k + 1) % 4)
                assert np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want[first * sout:], n, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "device")
    assert seen >= 40


def test_k_lind_streaming_store_flavour_on_aligned_magnified_rows():
    """Round 5: magnification on both axes into rows that start on 64-byte boundaries runs k_lind's flavour with streaming
    stores (k_lind_batch<0, IN, false, true>; dispatch::lind_nt): gamma-correct 32-bit and 24-bit sources, premul16 + gamma,
    premul16 below x1.2, into 32-bit and 24-bit rows whose strides are multiples of 64 bytes -- host road, and device buffers
    on 64-byte boundaries, whole images and row ranges."""
    cases = [(0, 640, 360, 0, 1280, 720, 1), (8, 640, 360, 0, 1280, 720, 1), (8, 640, 360, 8, 1280, 720, 1), (9, 500, 300, 8, 1024, 411, 1),
             (4, 640, 360, 4, 1280, 720, 1), (5, 500, 100, 6, 1008, 199, 1), (4, 1000, 400, 4, 1152, 461, 0), (0, 700, 300, 1, 768, 333, 1),
             (1, 320, 200, 9, 1024, 801, 1)]
    for k, (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in enumerate(cases):
        assert (w_out * H.bpp(t_out)) % 64 == 0
        assert sm.kernel_name(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)) == "k_lind_batch", (t_in, w_in, t_out, w_out)
        sin = w_in * H.bpp(t_in) + (k % 3)
        src = H.make_input(t_in, w_in, h_in, seed=70 + k, valid=bool(k & 1), stride=sin, pad=0)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin)
        got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin)
        assert np.array_equal(got, want), (t_in, w_in, h_in, t_out, w_out, h_out, "host")
        first, n = h_out // 4, h_out // 2
        rb = w_out * H.bpp(t_out)
        got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb, sin, first=first, n=n, src_misalign=k % 4, dst_misalign=0)
        assert np.array_equal(got, want[first * rb:(first + n) * rb]), (t_in, w_in, h_in, t_out, w_out, h_out, "device")


def test_magnification_into_unassociated_rows_through_k_up():
    """Round 5: k_up with unassociated 32-bit output from x3.2 -- un-premultiplied in registers through an LDS copy of the
    inverse table, the lane's four pixels as one 16-byte store: premultiplied and 24-bit sources into all four unassociated
    orders, aligned and odd widths, padded rows, partial tiles, row ranges, translucent and zero alpha."""
    shapes = [(480, 270, 1920, 1080), (333, 100, 1201, 361), (100, 40, 357, 143), (64, 20, 512, 161)]
    k = seen = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for t_in in (0, 1, 2, 3, 8, 9):
            for t_out in (4, 5, 6, 7):
                k += 1
                if (k % 2) and w_in * h_in > 60_000:
                    continue
                job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                assert not job.wide
                seen += sm.kernel_name(job) == "k_up_batch"
                sin = w_in * H.bpp(t_in) + 4 * (k % 3)
                sout = w_out * 4 + 4 * (k % 2)
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
                if t_in < 4 and k % 3 == 0:
                    src[::7] = 0                                   # patches of zero bytes: alpha 0 among them
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                rb = w_out * 4
                assert np.array_equal(H.rows_view(got, h_out, rb, sout), H.rows_view(want, h_out, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "host")
                first, n = h_out // 3, max(1, h_out // 2)
                got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n)
                assert np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want[first * sout:], n, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "device")
    assert seen >= 40


def test_24_bit_sources_into_32_bit_rows_two_columns_per_lane():
    """Round 5: k_tile takes two columns per lane for 24-bit sources behind one horizontal halving into dword-aligned 32-bit rows
    (a photograph into an RGBA buffer): both 24-bit orders into all eight 32-bit types at ratios 2-4, vertical minification and
    magnification, odd widths, padded rows, partial tiles; device buffers on dword boundaries and -- the one-column form -- off them."""
    shapes = [(4000, 300, 1600, 120), (3000, 200, 1421, 95), (1000, 64, 333, 200), (2000, 150, 515, 40), (700, 90, 260, 31)]
    k = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for t_in in (8, 9):
            for t_out in range(8):
                k += 1
                if (k % 3) and w_in * h_in > 500_000:
                    continue
                job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                assert not job.wide and sm.kernel_name(job) == "k_tile_batch"
                sin = w_in * 3 + (k % 4)
                sout = w_out * 4 + 4 * (k % 3)
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                rb = w_out * 4
                assert np.array_equal(H.rows_view(got, h_out, rb, sout), H.rows_view(want, h_out, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "host")
                first, n = h_out // 3, max(1, h_out // 2)
                for dmis in (0, 1 + k % 3):
                    got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n,
                                             src_misalign=k % 3, dst_misalign=dmis)
                    assert np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want[first * sout:], n, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, dmis)


def test_24_bit_rows_behind_one_halving_one_column_per_lane():
    """Round 5: into 24-bit rows k_tile goes back from four columns per lane to one (row groups of 3 KB) behind one halving on both
    axes for unassociated 32-bit sources, for premultiplied ones from ratio 3 and for 24-bit ones from ratio 3.7. Both sides of each
    threshold, every source class into both 24-bit orders, odd widths, padded rows, row ranges, misaligned device buffers."""
    shapes = [(4000, 300, 1600, 120), (4000, 300, 1300, 98), (4000, 300, 1053, 79), (3001, 200, 1421, 95), (2000, 150, 515, 40),
              (1900, 120, 611, 39)]
    k = 0
    for (w_in, h_in, w_out, h_out) in shapes:
        for t_in in (0, 1, 4, 5, 6, 8, 9):
            for t_out in (8, 9):
                k += 1
                if (k % 3) and w_in * h_in > 700_000:
                    continue
                bpp = 3 if t_in >= 8 else 4
                job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
                assert not job.wide and sm.kernel_name(job) == "k_tile_batch"
                sin = w_in * bpp + (k % 4 if bpp == 3 else 4 * (k % 2))
                sout = w_out * 3 + (k % 5)
                src = H.make_input(t_in, w_in, h_in, seed=k, valid=True, stride=sin, pad=0)
                want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                got = H.hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout)
                rb = w_out * 3
                assert np.array_equal(H.rows_view(got, h_out, rb, sout), H.rows_view(want, h_out, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, "host")
                first, n = h_out // 3, max(1, h_out // 2)
                for dmis in (0, 1 + k % 3):
                    got = H.hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, 0, sin, sout, first=first, n=n,
                                             src_misalign=(k % 3) if bpp == 3 else 0, dst_misalign=dmis)
                    assert np.array_equal(H.rows_view(got, n, rb, sout), H.rows_view(want[first * sout:], n, rb, sout)), (w_in, h_in, w_out, h_out, t_in, t_out, dmis)
