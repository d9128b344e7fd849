#!/usr/bin/env python3
"""64bpp class, box on both axes: the single-pass k_tile against the two-pass k_boxh + k_boxv by ratio and pixel size
(tuning build: SMOL_BOX2_MIN moves the threshold of box2_setup).
    SMOLSCALE_AMD_LIB=smolscale_amd/libsmolscale_amd_tuning.so python tools/box_threshold_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as B  # noqa: E402

for t, name in ((0, "RGBA8"), (8, "RGB8")):
    for w_out in (455, 410, 372, 341, 315, 293, 273, 256, 228, 205, 171, 146, 128):
        w_in = 4096 if t == 0 else 4000
        row = {"type": name, "ratio": round(w_in / w_out, 2)}
        for label, env in (("k_tile", "1000"), ("two_pass", "1")):
            os.environ["SMOL_BOX2_MIN"] = env
            B.CONFIGS["T"] = (t, w_in, w_in * 3 // 4, None, t, w_out, w_out * 3 // 4, 10, "sweep point")
            line = B.run_config("T", 10, 2, settle=0.1)
            row[label] = line["step_us"]
            row[label + "_kernel"] = line["kernel"][:12]
        print(json.dumps(row), flush=True)
