#!/usr/bin/env python3
"""Round 6: the headline over time -- the placed arenas of bench.py, four destination arenas in rotation (and one), launched back to back for
ten seconds in chunks of 200 launches, with the clocks, power and temperatures rocm-smi shows between chunks: what a device that has
just woken up (bursts: 325 us) and one under sustained load (333-341 us) differ in.
    python tools/exp_sustained.py [seconds]"""
import importlib.util, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
import smolscale_amd as sm
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
job = sm.plan_job(B.PIX_TYPE, B.W_IN, B.H_IN, B.PIX_TYPE, B.W_OUT, B.H_OUT, 0)
slot_in = (B.H_IN * B.W_IN * 4 + 64 + 255) // 256 * 256
slot_out = (B.H_OUT * B.W_OUT * 4 + 64 + 255) // 256 * 256
n = 32
src, dst, rot, report = B.pick_placement(dev, 0, n, slot_in, slot_out,
                                         lambda s, d: [B._image(job, s[f].data_ptr(), d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(n)])
print("placement: picked %.1f us, %d arenas" % (report["picked_us"], report["dst_rotation"]), flush=True)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        keep = []
        for l in out.splitlines():
            if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power (W)", "junction", "memory)")):
                keep.append(l.split(":", 1)[-1].strip().replace("clock level: ", "")[:60])
        return " | ".join(keep)
    except Exception as e:      # noqa: BLE001
        return str(e)[:60]


time.sleep(2.0)                 # an idle device
print("idle:", smi(), flush=True)
one = rot.batches[0]
t0 = time.time()
k = 0
while time.time() - t0 < seconds:
    us4 = rot.time(0, 200) / 200 * 1e3
    us1 = one.time(0, 40) / 40 * 1e3
    if k < 10 or k % 10 == 0:
        print("t=%5.2f s  four arenas %.1f us  one arena %.1f us   %s" % (time.time() - t0, us4, us1, smi() if k % 10 == 0 else ""), flush=True)
    k += 1
print("after an idle second:")
time.sleep(1.0)
for _ in range(5):
    print("   four arenas %.1f us (12 launches)" % (rot.time(0, 12) / 12 * 1e3))
