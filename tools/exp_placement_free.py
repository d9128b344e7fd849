#!/usr/bin/env python3
"""Round 6: bench.py's pick_destination () saw its picked candidate probe at 326 us and the timed region right after run at 333 us.
What between the two changes the speed -- destroying the other candidates' batches, freeing their arenas, emptying torch's cache?
Two source arenas x four destination candidates; the best pair is followed through each clean-up step.   python tools/exp_placement_free.py"""
import importlib.util, os, sys
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
srcs = [torch.randint(0, 256, (32, slot_in), dtype=torch.uint8, device=dev) for _ in range(2)]
def mk(src, d): return sm.HipBatch([B._image(job, src[f].data_ptr(), d[f].data_ptr(), B.W_IN * 4, B.W_OUT * 4, 0, B.H_OUT) for f in range(32)], device=0)
def t(b, n=16): return b.time(1, n) / n * 1e3
dsts, spacers, bs = [], [], {}
for k in range(4):
    d = torch.zeros((32, slot_out), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
    dsts.append(d)
    for i, s in enumerate(srcs):
        bs[(i, k)] = mk(s, d)
    if k == 0: bs[(0, 0)].time(0, 500)
    if k < 3: spacers.append(torch.empty(3 << 30, dtype=torch.uint8, device=dev))
tm = {key: t(b) for key, b in bs.items()}
print("probes  ", "  ".join("%d,%d=%.1f" % (i, k, v) for (i, k), v in sorted(tm.items())), flush=True)
best = min(tm, key=tm.get)
worst = max(tm, key=tm.get)
print("best %s %.1f   worst %s %.1f" % (best, tm[best], worst, tm[worst]))
for tag, key in (("best", best), ("worst", worst)):
    print("%-5s again                         %.1f" % (tag, t(bs[key])), flush=True)
for key, b in list(bs.items()):
    if key not in (best, worst):
        b.destroy(); del bs[key]
print("other batches destroyed:   best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
del spacers
print("spacers deleted:           best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
keep = {best[1], worst[1]}
dsts = [d if k in keep else None for k, d in enumerate(dsts)]
print("other dst arenas deleted:  best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
srcs2 = [s if i in (best[0], worst[0]) else None for i, s in enumerate(srcs)]
srcs = srcs2
print("other src arena deleted:   best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
torch.cuda.empty_cache()
print("cache emptied:             best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
bs[best].time(0, 600)
print("after 600 launches:        best %.1f  worst %.1f" % (t(bs[best]), t(bs[worst])), flush=True)
print("x20:                       best %.1f  worst %.1f" % (bs[best].time(5, 20) / 20 * 1e3, bs[worst].time(5, 20) / 20 * 1e3), flush=True)
