"""GPU parity over the reference's OWN benchmark matrix: the proportional sweep of bench.sh:39-43
-- 5-1920-1080-0.01-2-200, 5-3840-2160-0.01-2-200, 5-7680-4320-0.01-1-200, ARGB8 premultiplied in and out
(test.c:20) -- with the size rule of test.c:1030-1031. Every one of the 200 output sizes of the 1080p test and every
8th size of the 4K and 8K tests goes through the device-pointer entry at FULL width (these sizes select every k_tile /
k_boxh / k_up / k_pow2 geometry the sweep benchmark times); the whole image is produced on the GPU and three row strips
of it (top, middle, bottom) are compared byte for byte with the oracle, which computes just those rows."""
import numpy as np
import pytest

import helpers as H
import smolscale_amd as sm

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not H.have_gpu(), reason="needs a HIP device")]

ARGB_P = 2      # SMOL_PIXEL_ARGB8_PREMULTIPLIED


def steps_of(w_in, h_in, smin, smax, n):
    """test.c:1011-1031: sizes from scale_min to scale_max in n steps, clamped to 1..65535"""
    w0, w1 = int(w_in * smin), int(w_in * smax)
    h0, h1 = int(h_in * smin), int(h_in * smax)
    ws = (w1 - w0) / (n - 1.0)
    hs = (h1 - h0) / (n - 1.0)
    for s in range(n):
        yield (min(max(int(w0 + s * ws), 1), 65535), min(max(int(h0 + s * hs), 1), 65535))


def _strips(h_out):
    if h_out <= 40:
        return [(0, h_out)]
    mid = h_out // 2 - 8
    return [(0, 12), (mid, 16), (h_out - 12, 12)]


def _sweep(w_in, h_in, smax, every):
    import ctypes as C
    src = H.make_input(ARGB_P, w_in, h_in, seed=w_in, valid=True, pad=64)
    lib = sm.lib()
    dsrc = sm.DeviceBuffer(src.size + 64)
    dsrc.upload(src)
    sizes = list(steps_of(w_in, h_in, 0.01, smax, 200))[::every]
    ddst = sm.DeviceBuffer(max(w * h for w, h in sizes) * 4 + 64)
    kernels = set()
    hist = {}
    for (w_out, h_out) in sizes:
        job = H.plan_job(ARGB_P, w_in, h_in, ARGB_P, w_out, h_out, 0)
        plan = C.c_void_p()
        sm.check(lib.smol_hip_plan_create(C.byref(job), 0, C.byref(plan)), "plan_create")
        ddst.fill(0x5A)
        sm.check(lib.smol_hip_scale_rows_device(plan, dsrc.ptr, w_in * 4, ddst.ptr, w_out * 4, 0, h_out, None),
                 "scale_rows_device")
        sm.check(lib.smol_hip_device_synchronize(0), "sync")
        lib.smol_hip_plan_destroy(plan)
        kernels.add((job.filter_h, job.halvings_h, job.filter_v, job.halvings_v))
        kn = sm.kernel_name(job)
        hist[kn] = hist.get(kn, 0) + 1
        for first, n in _strips(h_out):
            want = H.oracle_scale(src, ARGB_P, w_in, h_in, ARGB_P, w_out, h_out, 0, None, None, first, n)
            got = ddst.download(n * w_out * 4, offset=first * w_out * 4)
            assert np.array_equal(got, want), (w_in, h_in, w_out, h_out, first, n)
    dsrc.free()
    ddst.free()
    print("kernel histogram %dx%d:" % (w_in, h_in),
          ", ".join("%s %d" % kv for kv in sorted(hist.items(), key=lambda kv: -kv[1])))    # pytest -s shows it
    # the correctness backstops (k_general / k_wavepx / k_wtile) take none of the reference's benchmark matrix
    assert not any(k.startswith(("k_general", "k_wavepx", "k_wtile")) for k in hist), hist
    return kernels


def test_reference_sweep_1080p_all_200_sizes():
    classes = _sweep(1920, 1080, 2.0, 1)
    # the sweep crosses every planner class: box, bilinear 2H / 1H / 0H down, copy-ish and magnification
    assert {f for (f, _, _, _) in classes} >= {2, 3}
    assert {h for (f, h, _, _) in classes if f == 2} == {0, 1, 2}


def test_reference_sweep_4k_every_8th_size():
    _sweep(3840, 2160, 2.0, 8)


def test_reference_sweep_8k_every_8th_size():
    _sweep(7680, 4320, 1.0, 8)
