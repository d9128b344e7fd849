#!/usr/bin/env python3
"""The reference's own proportional benchmark matrix (bench.sh:39-43, test.c:993-1053)
run against this library, emitted in the reference's text formats so that results
can be laid beside smolscale's own `results/resize-*-{samples,average}.txt`:

    samples : "<w_out> <h_out> <elapsed_s> <pps>"            (test.c:343-355)
    average : "<mean pps> <5th percentile> <95th percentile>" (test.c:357-382)
    pps     = (w_in*h_in + w_out*h_out) / elapsed_s          (test.c:219-224)

Tests: reps-w-h-scale_min-scale_max-steps = 5-1920-1080-0.01-2-200, 5-3840-2160-0.01-2-200,
5-7680-4320-0.01-1-200; ARGB8 premultiplied in and out, no sRGB (test.c:20, 785-807).
Rows per test:
    smol-hip        host pointers through smol_scale_simple (PCIe-inclusive, like the reference's
                    in-memory `smol' row)
    smol-hip-dev    device-resident input/output, one kernel per image, F DISTINCT source frames rotated
                    (F x frame bytes >= 600 MB, F >= 4: more than twice the 256 MB Infinity Cache, so the
                    row is an HBM number), hipEvents around the F kernels, time per image
    smol-hip-batch  the same F frames as ONE descriptor-table launch, time per image
    smol, smol-mt   the reference itself (oracle/_ref, AVX2 build; one thread / every host thread with the
                    strip pattern of test.c:838-883) -- only through tests/bench_sweep_with_reference.py:
                    tools/ never loads anything from oracle/

    python tools/bench_sweep.py [--out results] [--tests 5-1920-1080-0.01-2-200,...] [--steps N]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import smolscale_amd as sm  # noqa: E402

ARGB_P = sm.SMOL_PIXEL_ARGB8_PREMULTIPLIED
DEFAULT_TESTS = "5-1920-1080-0.01-2-200,5-3840-2160-0.01-2-200,5-7680-4320-0.01-1-200"


def steps_of(w_in, h_in, smin, smax, n):
    w0, w1 = int(w_in * smin), int(w_in * smax)
    h0, h1 = int(h_in * smin), int(h_in * smax)
    ws = (w1 - w0) / (n - 1.0) if n > 1 else 0.0
    hs = (h1 - h0) / (n - 1.0) if n > 1 else 0.0
    for s in range(n):
        yield (min(max(int(w0 + s * ws), 1), 65535), min(max(int(h0 + s * hs), 1), 65535))


def write_results(out_dir, test, name, samples):
    pps = sorted(p for (_, _, _, p) in samples)
    with open(os.path.join(out_dir, "resize-%s-samples.txt" % test), "a") as f:
        f.write('"%s"\n' % name)
        for (w, h, t, p) in samples:
            f.write("%u %u %.9f %.1f\n" % (w, h, t, p))     # (the reference prints microseconds; a kernel here takes a few)
        f.write("\n\n")
    with open(os.path.join(out_dir, "resize-%s-average.txt" % test), "a") as f:
        f.write('"%s" %.1f %.1f %.1f\n' % (name, sum(pps) / len(pps), pps[int(len(pps) * 0.05)], pps[int(len(pps) * 0.95)]))
    return sum(pps) / len(pps)


def main(argv=None, reference_rows=None):
    """reference_rows (optional): callable (src, w_in, h_in, sizes, reps) -> {row name: samples}"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "results"))
    ap.add_argument("--tests", default=DEFAULT_TESTS)
    ap.add_argument("--steps", type=int, default=0, help="override the number of steps (quick runs)")
    ap.add_argument("--pixel-type", type=int, default=ARGB_P, help="pixel type of both sides (default ARGB8 premultiplied, as the reference's matrix; 32-bit types only)")
    ap.add_argument("--srgb", type=int, default=0, help="with_srgb (the reference's matrix runs without)")
    ap.add_argument("--no-host-row", action="store_true", help="skip the host-pointer row (device rows only: quick looks)")
    ap.add_argument("--no-rotate", action="store_true", help="device rows into ONE output set per size (rounds 2-5's protocol)")
    ap.add_argument("--placement", action="store_true",
                    help="device rows: source frames at the head of ONE 104 GB region, outputs at the best of six positions in it (probed once "
                         "per test at half size): source and destination in different zones of the device's memory, the streaming kernels' "
                         "fast mode (profiles/r06_headline_modes.txt); default: separate allocations wherever the allocator puts them")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    rng = np.random.default_rng(7)

    for test in args.tests.split(","):
        reps, w_in, h_in, smin, smax, n = test.split("-")
        reps, w_in, h_in, n = int(reps), int(w_in), int(h_in), int(n)
        smin, smax = float(smin), float(smax)
        if args.steps:
            n = args.steps
        frame_bytes = w_in * h_in * 4
        n_frames = max(4, min(64, -(-600_000_000 // frame_bytes)))
        src = rng.integers(0, 256, frame_bytes, dtype=np.uint8)
        dsrcs = []
        region = out_base = None
        if args.placement:
            if not hasattr(main, "_region"):
                main._region = sm.DeviceBuffer(104 << 30)
            region = main._region

            class _Win:                                      # a window of the region with DeviceBuffer's interface
                def __init__(self, off, n):
                    self.off, self.ptr, self.n = off, region.ptr + off, n
                def upload(self, host):
                    region.upload(host, offset=self.off)
                def free(self):
                    pass
        pitch = (frame_bytes + 64 + 255) // 256 * 256
        for f in range(n_frames):
            d = _Win(f * pitch, frame_bytes + 64) if region is not None else sm.DeviceBuffer(frame_bytes + 64)
            d.upload(np.roll(src, 4 * f))                    # distinct contents, distinct addresses
            dsrcs.append(d)
        sizes = list(steps_of(w_in, h_in, smin, smax, n))
        if region is not None:
            # where the outputs go: the best of six positions, judged once per test on the half-size job
            out_max = max((w * h * 4 + 64 + 255) // 256 * 256 for (w, h) in sizes) * n_frames
            lo, hi = (n_frames * pitch + (2 << 20)) >> 21 << 21, (region.n_bytes - out_max - (2 << 20)) >> 21 << 21
            pw, ph = max(1, w_in // 2), max(1, h_in // 2)
            pjob = sm.plan_job(args.pixel_type, w_in, h_in, args.pixel_type, pw, ph, args.srgb)
            pslot = (pw * ph * 4 + 64 + 255) // 256 * 256
            best = None
            for base in [hi] + [(lo + (hi - lo) * k // 5) >> 21 << 21 for k in range(5)]:
                ims = []
                for f in range(n_frames):
                    im = sm.SmolHipImage()
                    im.job = pjob
                    im.src, im.dst = dsrcs[f].ptr, region.ptr + base + f * pslot
                    im.rowstride_in, im.rowstride_out = w_in * 4, pw * 4
                    im.first_row, im.n_rows = 0, ph
                    ims.append(im)
                b = sm.HipBatch(ims)
                b.time(8, 1)
                t = b.time(1, 8)
                b.destroy()
                if best is None or t < best[0]:
                    best = (t, base)
            out_base = best[1]
        host, dev, bat = [], [], []
        for (w_out, h_out) in sizes:
            dst = np.empty(w_out * h_out * 4, np.uint8)
            # host-pointer row: fastest of `reps' (benchmark_postprocess, test.c:319-340)
            best = 1e9
            for _ in range(0 if args.no_host_row else reps):
                t0 = time.perf_counter()
                sm.smol_scale_simple(src, args.pixel_type, w_in, h_in, w_in * 4, dst, args.pixel_type, w_out, h_out, w_out * 4, args.srgb)
                best = min(best, time.perf_counter() - t0)
            host.append((w_out, h_out, best, (w_in * h_in + w_out * h_out) / best))
            # device-resident rows: F frames, each with its own output
            out_slot = (dst.size + 64 + 255) // 256 * 256
            ddst = _Win(out_base, out_slot * n_frames) if region is not None else sm.DeviceBuffer(out_slot * n_frames)
            job = sm.plan_job(args.pixel_type, w_in, h_in, args.pixel_type, w_out, h_out, args.srgb)
            images = []
            for f in range(n_frames):
                im = sm.SmolHipImage()
                im.job = job
                im.src, im.dst = dsrcs[f].ptr, ddst.ptr + f * out_slot
                im.rowstride_in, im.rowstride_out = w_in * 4, w_out * 4
                im.first_row, im.n_rows = 0, h_out
                images.append(im)
            batch = sm.HipBatch(images)
            # (round 6) the one-launch row rotates over enough output sets that 512 MB of other output lie between two writes of a line
            # -- into ONE set the Infinity Cache absorbs the writes while the set stays below 256 MB, and the sweep showed a cliff where
            # 19 outputs pass that size (profiles/r06_dst_rotation.txt); --no-rotate: one set, as rounds 2-5
            n_rot = 1 if args.no_rotate else max(1, min(16, -(-(512 << 20) // (out_slot * n_frames))))
            extra_dst, rot = [], [batch]
            for k in range(1, n_rot):
                if region is not None:
                    base_k = out_base - k * ((out_slot * n_frames + (2 << 20)) >> 21 << 21)
                    if base_k < n_frames * pitch + (2 << 20):
                        break
                    dk_ptr = region.ptr + base_k
                else:
                    dk = sm.DeviceBuffer(out_slot * n_frames)
                    extra_dst.append(dk)
                    dk_ptr = dk.ptr
                for f, im in enumerate(images):
                    im.dst = dk_ptr + f * out_slot
                rot.append(sm.HipBatch(images))
            for f, im in enumerate(images):
                im.dst = ddst.ptr + f * out_slot
            timed = sm.HipBatchRotation(rot) if len(rot) > 1 else batch
            # The clock governor needs tens of milliseconds of load to settle and the host-pointer row above leaves the
            # GPU idle most of the time: rounds 2-3 timed 2 launches after 1 warm-up and read instruction-bound sizes
            # 20-30 % slow (ratio 1.1 at 4K: 0.55 of the roofline here, 0.79 in tools/bench_configs.py, which settles
            # for 0.3 s). Now: ~40 ms of warm-up launches, then the fastest of three groups of four (the reference
            # takes the fastest of its repetitions too: test.c:319-340).
            probe = timed.time(1, 2) / 2
            n_warm = max(2, min(400, int(40.0 / max(probe, 1e-3))))
            timed.time(n_warm, 1)
            t = min(timed.time(0, 4) / 4 for _ in range(3)) * 1e-3 / n_frames
            bat.append((w_out, h_out, t, (w_in * h_in + w_out * h_out) / t))
            timed.use_graph(True)                           # one kernel node per image
            t = min(timed.time(1, 4) / 4 for _ in range(2)) * 1e-3 / n_frames
            dev.append((w_out, h_out, t, (w_in * h_in + w_out * h_out) / t))
            for b in rot:
                b.destroy()
            for x in extra_dst:
                x.free()
            ddst.free()
        for d in dsrcs:
            d.free()
        means = {}
        if reference_rows is not None:
            for name, samples in reference_rows(src, w_in, h_in, sizes, reps).items():
                means[name] = write_results(args.out, test, name, samples)
        means["smol-hip"] = write_results(args.out, test, "smol-hip (host pointers, PCIe inclusive)", host)
        means["smol-hip-dev"] = write_results(args.out, test, "smol-hip-dev (device resident, %d frames rotated, one kernel per image)" % n_frames, dev)
        means["smol-hip-batch"] = write_results(args.out, test, "smol-hip-batch (device resident, %d frames in one launch)" % n_frames, bat)
        print("%s: mean pps " % test + ", ".join("%s %.3e" % kv for kv in means.items()), flush=True)


if __name__ == "__main__":
    main()
