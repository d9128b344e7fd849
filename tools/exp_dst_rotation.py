#!/usr/bin/env python3
"""Round 6: does the Infinity Cache (256 MB) absorb the headline's 128 MB of OUTPUT when the same batch is launched again and again?
The same 32 source frames into K destination arenas used in rotation (K x 128 MB: K = 1 is what bench.py times; K = 4 is 512 MB of
distinct output between two writes of the same line), all windows of one region far from the sources; wall time per launch over
200 launches on one stream, and torch events around them.      python tools/exp_dst_rotation.py"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
n = 32
src, dst, b0, rep = B.pick_placement(dev, 0, n, slot_in, slot_out,
                                     lambda sr, d: [B._image(job, sr[f].data_ptr(), d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(n)])
print("placement: src %s  dst %s" % (rep["src_windows_us"], rep["dst_windows_us"]))
region = B.placement_region(dev)
base = dst.data_ptr() - region.data_ptr()
KMAX = 8
dsts = []
for k in range(KMAX):
    off = base - k * (n * slot_out + (1 << 20))          # K arenas side by side, going down from the picked one
    dsts.append(region[off:off + n * slot_out].view(n, slot_out))
batches = [sm.HipBatch([B._image(job, src[f].data_ptr(), d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(n)], device=0) for d in dsts]
for b in batches:
    print("  arena alone: %.1f us per launch" % (b.time(3, 20) / 20 * 1e3))
stream = torch.cuda.Stream()
for K in (1, 2, 4, 8, 1):
    with torch.cuda.stream(stream):
        for i in range(40):
            batches[i % K].launch(stream.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        t0 = time.perf_counter()
        N = 200
        for i in range(N):
            batches[i % K].launch(stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / N * 1e6
    print("K = %d destination arenas in rotation (%4d MB of output between two writes of a line): %.1f us per launch by events, %.1f by the host clock"
          % (K, K * n * slot_out >> 20, e0.elapsed_time(e1) / N * 1e3, wall), flush=True)
