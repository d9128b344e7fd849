"""Everything that needs TWO DISTINCT GPUs (the reference's analogue is its worker fan-out, test.c:838-883):
row strips of one host image over distinct devices (page-locked and pageable), the source slab crossing xGMI
(hipMemcpyPeerAsync, with and without peer access), thumbnails dealt out by image, and bench.py at 2 ranks on 2 GPUs
with the RCCL barrier. Skipped on a one-GPU box -- the pool this repo is developed on has one GPU per box -- so that
the first multi-GPU lease is a measurement, not a debugging session (tools/scale_curve.sh)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
import smolscale_amd as sm


def _n_devices():
    try:
        return sm.lib().smol_hip_device_count()
    except Exception:           # noqa: BLE001
        return 0


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_n_devices() < 2, reason="needs two HIP devices")]

JOBS = [
    (0, 2048, 1536, 0, 512, 384, 0),        # k_pow2
    (4, 640, 360, 1, 1920, 1080, 0),        # k_up
    (8, 3001, 2001, 0, 1200, 2001, 0),      # vertical COPY, 24-bit in
    (0, 2000, 1500, 2, 150, 90, 0),         # two-pass box
    (5, 1000, 700, 6, 333, 250, 1),         # sRGB-linear
]


def _devices(n):
    nd = _n_devices()
    return [i % nd for i in range(n)]


@pytest.mark.parametrize("n_strips", [2, 3, 8])
def test_rows_over_distinct_devices_pageable_and_pinned(n_strips):
    for spec in JOBS:
        t_in, w_in, h_in, t_out, w_out, h_out, srgb = spec
        src = H.make_input(t_in, w_in, h_in, seed=3, pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        plans = [sm.HipPlan(job, d) for d in _devices(n_strips)]
        got = np.full(want.size, 0xA5, np.uint8)
        sm.scale_rows_host_multi(plans, src, w_in * H.bpp(t_in), got, w_out * H.bpp(t_out), 0, h_out)
        assert np.array_equal(got, want), ("pageable", spec, n_strips)
        pin_src, pin_dst = sm.PinnedArray(src.size), sm.PinnedArray(want.size)
        pin_src.array[:] = src
        pin_dst.array[:] = 0xA5
        sm.scale_rows_host_multi(plans, pin_src.array, w_in * H.bpp(t_in), pin_dst.array, w_out * H.bpp(t_out), 0, h_out)
        assert np.array_equal(pin_dst.array, want), ("pinned", spec, n_strips)
        pin_src.free()
        pin_dst.free()
        for p in plans:
            p.destroy()


def test_source_slab_crosses_xgmi():
    """the image lives on GPU 1, the plan on GPU 0 (and the other way round)"""
    for spec in JOBS:
        t_in, w_in, h_in, t_out, w_out, h_out, srgb = spec
        src = H.make_input(t_in, w_in, h_in, seed=6, pad=16)
        want = H.oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        for (dev_src, dev_plan) in ((1, 0), (0, 1)):
            dsrc = sm.DeviceBuffer(src.size + 64, dev_src)
            dsrc.upload(src)
            plan = sm.HipPlan(H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb), dev_plan)
            first, n = h_out // 3, h_out - h_out // 3
            got = np.zeros(n * w_out * H.bpp(t_out), np.uint8)
            plan.scale_rows_peer(dev_src, dsrc.ptr, w_in * H.bpp(t_in), got, w_out * H.bpp(t_out), first, n)
            assert np.array_equal(got, want[first * w_out * H.bpp(t_out):]), (spec, dev_src, dev_plan)
            plan.destroy()
            dsrc.free()


def test_thumbnails_dealt_out_by_image_over_distinct_devices():
    rng = random.Random(77)
    devices = list(range(min(_n_devices(), 8)))
    n_img = 16 * len(devices)
    specs, images, dbufs, wants = [], [], [], []
    for i in range(n_img):
        d = devices[i % len(devices)]          # image i lives where it will be scaled: nothing crosses the fabric
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        src = H.make_input(t_in, 256, 256, seed=100 + i, valid=True, pad=16)
        dsrc, ddst = sm.DeviceBuffer(src.size, d), sm.DeviceBuffer(64 * 64 * H.bpp(t_out), d)
        dsrc.upload(src)
        im = sm.SmolHipImage()
        im.job = H.plan_job(t_in, 256, 256, t_out, 64, 64, 0)
        im.src, im.dst = dsrc.ptr, ddst.ptr
        im.rowstride_in, im.rowstride_out = 256 * H.bpp(t_in), 64 * H.bpp(t_out)
        im.first_row, im.n_rows = 0, 64
        specs.append((t_in, t_out))
        images.append(im)
        dbufs.append((dsrc, ddst))
        wants.append(H.oracle_scale(src, t_in, 256, 256, t_out, 64, 64))
    mb = sm.HipBatchMulti(images, devices)
    mb.launch()
    mb.sync()
    for want, (_, ddst), spec in zip(wants, dbufs, specs):
        assert np.array_equal(ddst.download(), want), spec
    mb.destroy()
    for a, b in dbufs:
        a.free()
        b.free()


def _last_json_line(out):
    for ln in reversed(out.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError(out[-2000:])


@pytest.mark.parametrize("config", ["c2", "c4", "c5mix"])
def test_bench_two_ranks_two_gpus_rccl_barrier(config):
    """one process per GPU under torch.distributed.run, the RCCL ("nccl") group as the timing barrier -- started
    as a child process before anything here has touched the GPU with an exec"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(H.ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "2", "--backend", "nccl", "--config", config]
    line = _last_json_line(subprocess.check_output(cmd, text=True, stderr=subprocess.DEVNULL, timeout=900, env=env))
    assert line["n_gpus"] == 2 and line["barrier_backend"] == "nccl" and line["value"] > 0
    assert line["verified_vs_oracle"]["all_ranks_ok"] is True


def test_public_api_over_all_devices_env():
    """SMOL_HIP_DEVICES=all: smol_scale_batch() splits its rows over every GPU (smolscale.h:71-73 semantics kept)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, helpers as H\n"
            "src = H.make_input(0, 2048, 1536, seed=9, pad=16)\n"
            "a = H.hip_scale(src, 0, 2048, 1536, 1, 700, 500)\n"
            "b = H.oracle_scale(src, 0, 2048, 1536, 1, 700, 500)\n"
            "print('EQUAL' if np.array_equal(a, b) else 'DIFFERENT')\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ, SMOL_HIP_DEVICES="all")
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=env, timeout=600)
    assert "EQUAL" in out
