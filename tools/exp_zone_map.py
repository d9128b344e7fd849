#!/usr/bin/env python3
"""Round 6: a map of the zones of device memory -- ONE region of most of the device (default 240 GB), cut into windows of 2 GB:
  (0) what the library's own probe says (smol_hip_memory_zones: two read streams side by side, pairs of windows)
  (1) a pure streaming READ and a pure streaming WRITE of every window (tools/microbench/zone_probe.hip): is a zone a property of the
      memory itself, and of which direction? Then two read streams (every window beside a fixed one), a plain copy and a 4:1
      reduction from every window into a fixed one
  (2) the headline's 2 GB source arena in every window, timed against a 128 MB destination arena at several fixed places
  (3) the headline's destination arena (four in rotation, 512 MB: every output byte leaves the cache) in every window, against a
      source in a fast and in a slow window
'#' = slower than the middle between the fastest and the slowest window of the row.
    python tools/exp_zone_map.py [region GB] [step MB]"""
import ctypes, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
step = (int(sys.argv[2]) if len(sys.argv) > 2 else 2048) << 20
zp = ctypes.CDLL(os.path.join(ROOT, "tools/microbench/libzone_probe.so"))
zp.smol_zone_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
n = 32
region = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
total = region.numel()
base = region.data_ptr()
src_bytes, dst_bytes = n * slot_in, n * slot_out
print("region %#x, %d GB, windows of %d MB" % (base, gb, step >> 20), flush=True)


def show(name, row):
    v = [t for t in row if t is not None]
    mid = (min(v) + max(v)) / 2 if max(v) > 1.02 * min(v) else max(v) * 2
    print("%-46s %s" % (name, "".join("x" if t is None else ("#" if t > mid else ".") for t in row)), flush=True)
    print("    us (%.0f..%.0f): " % (min(v), max(v)) + " ".join("--" if t is None else "%.0f" % t for t in row), flush=True)


def pure(off, nbytes, mode):
    us = ctypes.c_double()
    assert zp.smol_zone_probe(base + off, nbytes, mode, 2, 6, ctypes.byref(us)) == 0
    return us.value


zp.smol_zone_probe_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]


def pair(src_off, dst_off, dst_nbytes, reduce4, iters=6):
    us = ctypes.c_double()
    assert zp.smol_zone_probe_pair(base + src_off, base + dst_off, dst_nbytes, reduce4, 2, iters, ctypes.byref(us)) == 0
    return us.value


def headline(src_off, dst_off, rot=1, settle=False):
    bs = []
    for k in range(rot):
        ims = [B._image(job, base + src_off + f * slot_in, base + dst_off + k * dst_bytes + f * slot_out, B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(n)]
        bs.append(sm.HipBatch(ims, device=0))
    r = sm.HipBatchRotation(bs)
    if settle:
        r.time(0, 500)
    r.time(2, 4)
    t = r.time(0, 12) / 12 * 1e3
    for b in bs:
        b.destroy()
    return t


headline(0, total - dst_bytes, 1, True)
wins = list(range(0, total - src_bytes + 1, step))
import time
for w in (step, 512 << 20):
    t0 = time.time()
    z, kinds, c = sm.memory_zones(base, total, w)
    print("%-46s %s" % ("smol_hip_memory_zones, windows of %d MB" % (w >> 20), "".join("%d" % v for v in z[::max(1, step // w)])), flush=True)
    print("    %d kinds (GB of each: %s), contrast %.3f, %d windows in %.2f s; label changes at GB: %s" % (kinds,
          " ".join("%g" % (z.count(k) * w / 2**30) for k in range(kinds)), c, len(z), time.time() - t0,
          " ".join("%g" % (i * w / 2**30) for i in range(1, len(z)) if z[i] != z[i - 1])), flush=True)
if os.environ.get("ZONE_MAP_LIBRARY_ONLY", "0") == "1":
    sys.exit(0)
show("pure read of 2 GB", [pure(o, 2 << 30, 0) for o in wins])
show("pure write of 2 GB", [pure(o, 2 << 30, 1) for o in wins])
show("pure read of 2 GB (again)", [pure(o, 2 << 30, 0) for o in wins])
if os.environ.get("ZONE_MAP_READ2", "1") == "1":
    for ref in (0, 64 << 30, 150 << 30):
        for nbytes in (1 << 30, 256 << 20):
            show("two read streams of %d MB: every window and the one at %d GB" % (nbytes >> 20, ref >> 30),
                 [None if (o < ref + nbytes and ref < o + nbytes) else pair(o, ref, nbytes, 2, 6) for o in wins])
if os.environ.get("ZONE_MAP_COPIES", "1") == "1":
    for dst_off in (0, 64 << 30):
        for nbytes, red in ((1 << 30, 0), (64 << 20, 0), (512 << 20, 1)):
            show("%s of %d MB from every window to %d GB" % ("4:1 reduction" if red else "copy", (nbytes * (4 if red else 1)) >> 20, dst_off >> 30),
                 [None if (o < dst_off + nbytes and dst_off < o + nbytes * (4 if red else 1)) else pair(o, dst_off, nbytes, red, 6 if nbytes >= (256 << 20) else 20) for o in wins])
for name, dst_off in (("end", (total - dst_bytes) >> 12 << 12), ("at 64 GB", 64 << 30), ("at 150 GB", 150 << 30)):
    if dst_off + dst_bytes > total:
        continue
    show("source in every window, destination %s" % name,
         [None if (o < dst_off + dst_bytes and dst_off < o + src_bytes) else headline(o, dst_off) for o in wins])
for name, src_off in (("at 64 GB", 64 << 30), ("at 0", 0), ("at 118 GB", 118 << 30)):
    if src_off + src_bytes > total:
        continue
    show("4 destinations in every window, source %s" % name,
         [None if (o < src_off + src_bytes and src_off < o + 4 * dst_bytes) else headline(src_off, o, 4) for o in wins])
