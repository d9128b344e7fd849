"""Multi-rank row sharding on CPU: world_size 2 (and 3, uneven) over gloo.

Each rank holds ONLY the slab of input rows its output strip depends on, addresses
it through a virtual row-0 pointer exactly as bench.py / the multi-GPU scatter do,
produces its strip with the (host-emulated) kernel bodies, and the strips are
gathered with torch.distributed. The result must equal the oracle's one-shot
output: there is no exchange step on the data path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
import smolscale_amd as sm

JOBS = [
    # t_in, w_in, h_in, t_out, w_out, h_out, srgb
    (0, 256, 256, 0, 64, 64, 0),          # exact 4:1 (k_pow2 class)
    (4, 97, 61, 1, 150, 83, 0),           # magnify, unassoc -> premul
    (8, 300, 200, 2, 33, 17, 0),          # box x 2H... (300/33 > 8 -> box), RGB -> ARGB
    (0, 90, 47, 0, 90, 5, 1),             # copy x box, sRGB (wide)
    (8, 4095, 40, 0, 512, 40, 0),         # BASELINE config 4's classes (24-bit in, BILINEAR_2H x COPY): disjoint slabs
]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in JOBS:
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        first, n, r0, r1 = sm.shard(job, rank, world)
        stride_in = w_in * H.bpp(t_in)
        stride_out = w_out * H.bpp(t_out)
        rows_max = (h_out + world - 1) // world
        strip = np.zeros(rows_max * stride_out, np.uint8)
        if n:
            full = H.make_input(t_in, w_in, h_in, seed=11, valid=True, pad=16)
            # keep only this rank's slab; everything else is poisoned
            slab = H.aligned_copy(full[r0 * stride_in:(r1 + 1) * stride_in + 16])
            virt = slab.ctypes.data - r0 * stride_in
            import ctypes
            k = ctypes.c_int(0)
            out = np.zeros(n * stride_out + 16, np.uint8)
            rc = H.emu().emu_scale_rows(ctypes.byref(job), virt, stride_in, out.ctypes.data, stride_out,
                                        first, n, 0, ctypes.byref(k))
            assert rc == 0
            strip[:n * stride_out] = out[:n * stride_out]
        gathered = [torch.zeros(rows_max * stride_out, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(strip))
        if rank == 0:
            got = torch.cat(gathered).numpy()[:h_out * stride_out]
            full = H.make_input(t_in, w_in, h_in, seed=11, valid=True, pad=16)
            want = H.oracle_scale(full, t_in, w_in, h_in, t_out, w_out, h_out, srgb)
            ok = ok and bool(np.array_equal(got, want))
    if rank == 0:
        results.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_strips_gathered_over_gloo_equal_one_shot(world):
    H.emu()      # build the emulation library once, before forking workers
    ctx = mp.get_context("spawn")
    results = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert results.get() is True


def _thumb_specs(n_total):
    rng = np.random.default_rng(5)
    return [(int(a), int(b)) for a, b in rng.integers(0, 10, size=(n_total, 2))]


def _worker_images(rank, world, port, results):
    """BASELINE config 5's multi-GPU form: image i belongs to rank i % world, no image is
    split, nothing is exchanged; outputs are gathered only to be checked."""
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total, w_in, w_out = 14, 64, 16
    specs = _thumb_specs(n_total)
    slot = w_out * w_out * 4
    per_rank = (n_total + world - 1) // world
    mine = np.zeros(per_rank * slot, np.uint8)
    for k, i in enumerate(range(rank, n_total, world)):
        t_in, t_out = specs[i]
        src = H.make_input(t_in, w_in, w_in, seed=i, valid=True, pad=16)
        job = H.plan_job(t_in, w_in, w_in, t_out, w_out, w_out, 0)
        out = np.zeros(slot + 16, np.uint8)
        kc = ctypes.c_int(0)
        rc = H.emu().emu_scale_rows(ctypes.byref(job), src.ctypes.data, w_in * H.bpp(t_in), out.ctypes.data,
                                    w_out * H.bpp(t_out), 0, w_out, 0, ctypes.byref(kc))
        assert rc == 0
        mine[k * slot:(k + 1) * slot] = out[:slot]
    gathered = [torch.zeros(per_rank * slot, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(mine))
    if rank == 0:
        ok = True
        for i, (t_in, t_out) in enumerate(specs):
            src = H.make_input(t_in, w_in, w_in, seed=i, valid=True, pad=16)
            want = H.oracle_scale(src, t_in, w_in, w_in, t_out, w_out, w_out)
            got = gathered[i % world].numpy()[(i // world) * slot:(i // world) * slot + want.size]
            ok = ok and bool(np.array_equal(got, want))
        results.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_thumbnails_sharded_by_image_over_gloo(world):
    H.emu()
    ctx = mp.get_context("spawn")
    results = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_images, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert results.get() is True


def test_more_ranks_than_rows_gives_empty_strips():
    job = H.plan_job(0, 64, 8, 0, 32, 3, 0)
    got = [sm.shard(job, r, 8) for r in range(8)]
    assert sum(n for _, n, _, _ in got) == 3
    assert all(n == 0 and (r0, r1) == (0, -1) for _, n, r0, r1 in got[3:])


def test_strip_rule_and_slabs_cover_everything():
    for world in (1, 2, 3, 4, 7, 8):
        for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in JOBS + [(0, 4096, 4096, 0, 1024, 1024, 0),
                                                                     (8, 65535, 4096, 0, 8192, 4096, 0)]:
            job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
            covered = 0
            for r in range(world):
                first, n, r0, r1 = sm.shard(job, r, world)
                assert first == covered or n == 0
                covered += n
                if n:
                    assert 0 <= r0 <= r1 < h_in
            assert covered == h_out
