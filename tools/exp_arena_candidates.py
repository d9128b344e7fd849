#!/usr/bin/env python3
"""The headline's two per-process modes (profiles/r05_headline_modes.txt), next question: do arenas that are ALIVE TOGETHER in one
process -- so that they cannot be the same physical pages -- all run in that process's mode? K source arenas of 2 GB (+ 128 MB
destination each), built one after the other and all kept, the headline launch timed on each in turn, twice.
    python tools/exp_arena_candidates.py [K]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
import torch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
works = [B.work_c2(0, 1, dev, 0, 32) for _ in range(K)]
works[0].batch.time(0, 600)                      # settle the clock
for rep in range(2):
    for k, w in enumerate(works):
        src, dst = w.keep
        t = [w.batch.time(2, 40) / 40 * 1e3 for _ in range(2)]
        print("pid %d rep %d arena %d src %#x dst %#x  us per launch %s" % (os.getpid(), rep, k, src.data_ptr(), dst.data_ptr(),
                                                                         " ".join("%.1f" % x for x in t)), flush=True)
