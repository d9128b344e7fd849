#!/usr/bin/env python3
"""Does the headline's per-process mode (profiles/r05_headline_modes.txt) follow the STREAM (hardware queue) the launches go to?
One process, one arena; the same batch launched on sixteen different streams (torch streams: the runtime maps them onto its
hardware queues round-robin), 40 launches each, timed with events on that stream.    python tools/exp_streams.py"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = B.work_c2(0, 1, dev, 0, 32)
w.batch.time(0, 600)
print("library's own stream: %.1f us per launch" % (w.batch.time(0, 50) / 50 * 1e3))
streams = [torch.cuda.Stream(device=dev, priority=p) for p in (0,) * 12 + (-1,) * 4]
for k, st in enumerate(streams):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        for _ in range(5):
            w.batch.launch(st.cuda_stream)
        a.record(st)
        for _ in range(40):
            w.batch.launch(st.cuda_stream)
        b.record(st)
    st.synchronize()
    print("stream %2d (%#x, priority %d): %.1f us per launch" % (k, st.cuda_stream, st.priority, a.elapsed_time(b) / 40 * 1e3), flush=True)
print("library's own stream again: %.1f us per launch" % (w.batch.time(0, 50) / 50 * 1e3))
