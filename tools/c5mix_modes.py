#!/usr/bin/env python3
"""C5mix (10 000 mixed-type thumbnails, one launch) has been timing in two modes, 396-413 or 456-471 us, between processes
on one binary (profiles/r03i_perm_ab.txt). This runs the config several times in ONE process, every time on freshly
allocated arenas (a perturbing allocation of another size in between moves the addresses), and times each allocation
three times over: is the mode a property of the ALLOCATION (its addresses / physical placement) or of the process?

    python tools/c5mix_modes.py [rounds=8] [config=C5mix]
Under rocprofv3 --pmc FETCH_SIZE (or TCC_HIT_sum TCC_MISS_sum) the per-dispatch counters show what differs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import smolscale_amd as sm  # noqa: E402
import bench_configs as B  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "C5mix"
os.environ["SMOL_BENCH_PRINT_ARENA"] = "1"
rng = np.random.default_rng(3)
keep = []
for r in range(rounds):
    line = B.run_config(cfg, 20, 3, settle=0.2)
    print(json.dumps({"round": r, "step_us": line["step_us"], "min": line["step_us_min"], "median": line["step_us_median"],
                      "frac": line["frac_of_8TBps"], "probe_ns": line.get("arena_probe_ns"), "arena": line.get("arena_in_ptr")}), flush=True)
    # move the next round's arenas: hold on to an allocation of another size (freed two rounds later)
    keep.append(sm.DeviceBuffer(int(rng.integers(1, 64)) << 20))
    if len(keep) > 2:
        keep.pop(0).free()
