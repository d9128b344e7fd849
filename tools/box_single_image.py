#!/usr/bin/env python3
"""64bpp-class box filter: the single-pass k_tile against the two passes for 1, 2 and 4 images of 4096x3072 per launch (one
allocation per image) at ratios 9-24 (tuning build: SMOL_BOX2_MIN 1000 = always k_tile, 1 = always the two passes).
    SMOLSCALE_AMD_LIB=smolscale_amd/libsmolscale_amd_tuning.so python tools/box_single_image.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as B  # noqa: E402

for frames in (1, 2, 4):
    for w_out in (455, 390, 341, 256, 205, 171):
        row = {"frames": frames, "ratio": round(4096 / w_out, 1)}
        for label, env in (("k_tile", "1000"), ("two_pass", "1")):
            os.environ["SMOL_BOX2_MIN"] = env
            B.CONFIGS["T"] = (0, 4096, 3072, None, 0, w_out, w_out * 3 // 4, frames, "x")
            row[label] = B.run_config("T", 20, 3, settle=0.1, no_arena=True)["step_us"]
        print(json.dumps(row), flush=True)
