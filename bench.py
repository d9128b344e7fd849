#!/usr/bin/env python3
"""Headline benchmark: 4096x4096 -> 1024x1024 RGBA8 premultiplied downscale
(BASELINE.json configs[1]; the reference picks BILINEAR_1H on both axes for it).

One "step" = one pass of the hot path over one batch of synthetic frames that are
already resident in HBM: a single smol_hip_batch_launch() of 8*N distinct frames,
where N = number of ranks and rank r produces output-row strip r of every frame
(row ranges shard with no exchange). Per-rank bytes are therefore the
same at every N (weak scaling) and at N = 1 the batch is 8 whole frames = 570 MB
of algorithmic traffic per launch -- more than the 256 MiB Infinity Cache, so the
input really streams from HBM.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0. PyTorch is plumbing only (device memory, barrier).
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import smolscale_amd as sm  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
W_IN = H_IN = 4096
W_OUT = H_OUT = 1024
FRAMES_PER_GPU = 8
PIX_TYPE = sm.SMOL_PIXEL_RGBA8_PREMULTIPLIED


def cpu_baseline(seconds_budget=12.0):
    """The REAL reference (oracle/_ref, AVX2 build, all host cores, the reference's
    own strip-per-thread pattern test.c:838-883) on a bounded sample of the same
    workload; falls back to the oracle port when oracle/_ref is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H

    n_threads = os.cpu_count() or 1
    src = H.make_input(PIX_TYPE, W_IN, H_IN, seed=0, valid=True, pad=16)
    dst = np.zeros(W_OUT * H_OUT * 4 + 16, np.uint8)
    info = {}

    def run_ref(kind, threads, budget):
        """Calibrate on a few frames, then time one pthread run sized to ~budget seconds."""
        lib = H.ref(kind)
        lib.ref_probe_mt_time.restype = ctypes.c_double
        lib.ref_probe_mt_time.argtypes = [ctypes.c_void_p] + [ctypes.c_uint32] * 4 + \
                                         [ctypes.c_void_p] + [ctypes.c_uint32] * 4 + [ctypes.c_uint32] * 3

        def go(n_frames):
            return lib.ref_probe_mt_time(src.ctypes.data, PIX_TYPE, W_IN, H_IN, W_IN * 4,
                                         dst.ctypes.data, PIX_TYPE, W_OUT, H_OUT, W_OUT * 4, 0, threads, n_frames)

        t_cal = go(4)
        n = int(max(8, min(20000, budget / max(t_cal / 4, 1e-6))))
        total = go(n)
        return n, total

    if H.have_ref("avx2") and H.have_ref("generic"):
        n, total = run_ref("avx2", n_threads, seconds_budget * 0.5)
        info = {
            "value": round(n * W_OUT * H_OUT / total / 1e6, 1), "unit": "Mpix/s",
            "cores": n_threads, "kind": "reference",
            "sample": "%d frames 4096x4096->1024x1024 RGBA8 premul through the reference's AVX2 backend "
                      "(oracle/_ref, gcc -O2 -mavx2): %d pthreads, one output-row strip each via smol_scale_batch "
                      "(test.c:838-883 pattern), %.1f s wall" % (n, n_threads, total),
        }
        n1, total1 = run_ref("generic", 1, seconds_budget * 0.2)
        info["generic_1thread_Mpix_s"] = round(n1 * W_OUT * H_OUT / total1 / 1e6, 1)
        n2, total2 = run_ref("avx2", 1, seconds_budget * 0.15)
        info["avx2_1thread_Mpix_s"] = round(n2 * W_OUT * H_OUT / total2 / 1e6, 1)
        n3, total3 = run_ref("generic", n_threads, seconds_budget * 0.15)
        info["generic_allthreads_Mpix_s"] = round(n3 * W_OUT * H_OUT / total3 / 1e6, 1)
        want = dst[:W_OUT * H_OUT * 4].copy()
    else:
        t0 = time.perf_counter()
        n = 0
        while True:
            want = H.oracle_scale(src, PIX_TYPE, W_IN, H_IN, PIX_TYPE, W_OUT, H_OUT)
            n += 1
            if time.perf_counter() - t0 > seconds_budget or n >= 100:
                break
        total = time.perf_counter() - t0
        info = {
            "value": round(n * W_OUT * H_OUT / total / 1e6, 1), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "%d frames 4096x4096->1024x1024 through the scalar C oracle (oracle/smol_oracle.c), %.1f s" % (n, total),
        }
    return info, src, want


# The other BASELINE.json configs, single GPU, device resident: not the metric (that is config 2
# above) -- reported beside it so that the driver's own run records them. name: (type_in, w_in,
# h_in, stride_in, type_out, w_out, h_out, images per launch)
OTHER_CONFIGS = {
    "C1 640x480->320x240 RGBA8 premul x384": (0, 640, 480, None, 0, 320, 240, 384),
    "C3 1920x1080 RGBA unassociated->7680x4320 RGBA premul x4": (4, 1920, 1080, None, 0, 7680, 4320, 4),
    "C4 65535x4096 RGB8 (stride 196605)->8192x4096 RGBA8 x1": (8, 65535, 4096, 65535 * 3, 0, 8192, 4096, 1),
    "C5 256x256->64x64 RGBA8 premul x10000": (0, 256, 256, None, 0, 64, 64, 10000),
}


def other_configs(dev, local_rank):
    """Time the other BASELINE configs with synthetic device-resident data (hipEvents, after
    0.2 s of warm-up load each). Any failure is reported in place of the numbers."""
    out = {}
    for name, (t_in, w_in, h_in, stride_in, t_out, w_out, h_out, n_img) in OTHER_CONFIGS.items():
        try:
            bpp_in, bpp_out = sm.bytes_per_pixel(t_in), sm.bytes_per_pixel(t_out)
            s_in, s_out = stride_in or w_in * bpp_in, w_out * bpp_out
            src = torch.randint(0, 256, (n_img, h_in * s_in + 64), dtype=torch.uint8, device=dev)
            dst = torch.zeros((n_img, h_out * s_out + 64), dtype=torch.uint8, device=dev)
            job = sm.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
            images = []
            for i in range(n_img):
                im = sm.SmolHipImage()
                im.job = job
                im.src, im.dst = src[i].data_ptr(), dst[i].data_ptr()
                im.rowstride_in, im.rowstride_out = s_in, s_out
                im.first_row, im.n_rows = 0, h_out
                images.append(im)
            torch.cuda.synchronize()
            b = sm.HipBatch(images, device=local_rank)
            n_launches, kname = b.info()
            probe = b.time(2, 5) / 5
            ms = b.time(int(200.0 / max(probe, 1e-3)), 20) / 20
            algo = n_img * (w_in * h_in * bpp_in + w_out * h_out * bpp_out)
            gbs = algo / (ms * 1e-3) / 1e9
            out[name] = {"kernel": kname, "launches": n_launches, "step_us": round(ms * 1e3, 1),
                         "GBps": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                         "Mpix_out_per_s": round(n_img * w_out * h_out / (ms * 1e-3) / 1e6, 1)}
            b.destroy()
            del src, dst
        except Exception as e:                      # noqa: BLE001 -- a side table must not break the bench line
            out[name] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side table of the other BASELINE configs (N=1 only, a few seconds)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the timing barrier (nccl = RCCL)")
    ap.add_argument("--share-devices", action="store_true",
                    help="testing aid: rank r uses GPU r %% device_count (several ranks on one GPU; needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the product has no CPU path to measure")

    if args.share_devices:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- the job and this rank's shard ------------------------------------------------
    job = sm.plan_job(PIX_TYPE, W_IN, H_IN, PIX_TYPE, W_OUT, H_OUT, 0)
    first_row, n_rows, r0, r1 = sm.shard(job, rank, world)
    n_frames = FRAMES_PER_GPU * world
    in_stride, out_stride = W_IN * 4, W_OUT * 4
    slab_rows = r1 - r0 + 1

    gen = torch.Generator(device=dev)
    gen.manual_seed(0x9E3779B9 + rank)
    src = torch.randint(0, 256, (n_frames, slab_rows, in_stride), dtype=torch.uint8, device=dev, generator=gen)
    dst = torch.zeros((n_frames, n_rows, out_stride), dtype=torch.uint8, device=dev)

    # rank 0 at N=1 replaces frame 0 by the CPU baseline's seeded frame so the GPU
    # result can be checked against the reference's bytes in the same run
    cpu = None
    want = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, frame0, want = cpu_baseline()
        src[0].copy_(torch.from_numpy(frame0[:H_IN * in_stride].reshape(H_IN, in_stride)))

    images = []
    for f in range(n_frames):
        im = sm.SmolHipImage()
        im.job = job
        # address of (virtual) input row 0: the slab starts at row r0
        im.src = src[f].data_ptr() - r0 * in_stride
        im.dst = dst[f].data_ptr()
        im.rowstride_in, im.rowstride_out = in_stride, out_stride
        im.first_row, im.n_rows = first_row, n_rows
        images.append(im)
    torch.cuda.synchronize()
    batch = sm.HipBatch(images, device=local_rank)
    n_launches, kernel_name = batch.info()

    # ---- warm-up, then exactly K timed steps --------------------------------------------
    if args.warmup:
        batch.time(0, args.warmup)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = batch.time(0, args.steps)          # K launches bracketed by hipEvents on their stream
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    verified = None
    if want is not None:
        got = dst[0].cpu().numpy().reshape(-1)
        verified = bool(np.array_equal(got, want))

    if rank == 0:
        out_pix_per_step = n_frames * W_OUT * H_OUT                       # all ranks together
        in_pix_per_step = n_frames * W_IN * H_IN
        value = out_pix_per_step * args.steps / elapsed / 1e6
        # algorithmic bytes ONE launch (this rank) moves: every source pixel of the
        # strip read once, every destination pixel written once
        src_rows_algo = n_rows * (H_IN // H_OUT)
        bytes_per_launch = n_frames * (src_rows_algo * W_IN * 4 + n_rows * W_OUT * 4)
        launch_ms = kernel_ms / args.steps / max(1, n_launches)
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/sec 4K->1K RGBA8 downscale (output pixels)",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "4096x4096->1024x1024 RGBA8 premultiplied (reference filter: BILINEAR_1H x BILINEAR_1H), "
                            "batch of %d distinct frames per step resident in HBM, output-row strips sharded over %d rank(s)"
                            % (n_frames, world),
                "frames_per_step": n_frames,
                "rows_per_rank": n_rows,
                "launches_per_step": n_launches,
            },
            "ref_style_pps_Mpix_s": round((out_pix_per_step + in_pix_per_step) * args.steps / elapsed / 1e6, 1),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "bytes_per_launch": bytes_per_launch,
                "launch_us": round(launch_ms * 1e3, 3),
            },
        }
        if world == 1 and not args.no_other_configs:
            line["other_baseline_configs"] = other_configs(dev, local_rank)
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if verified is not None:
            line["bit_exact_vs_reference"] = verified
        print(json.dumps(line), flush=True)

    batch.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
