#!/usr/bin/env python3
"""One image per launch against eight: where a kernel's grid is too small for one image, the single image costs far more
than an eighth of the batch. 4096x3072 sources, device resident (tools/bench_configs.py machinery), step us.
    python tools/single_image_survey.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as B  # noqa: E402

CASES = [
    ("4:1 RGBA", 0, 0, 0, 1024, 768), ("ratio 1.5 RGBA", 0, 0, 0, 2731, 2048), ("ratio 3 RGBA", 0, 0, 0, 1365, 1024),
    ("ratio 7 RGBA", 0, 0, 0, 585, 439), ("x1.5 RGBA", 0, 0, 0, 6144, 4608), ("x2 RGBA", 0, 0, 0, 8192, 6144),
    ("ratio 2.5 RGB8", 8, 8, 0, 1638, 1229), ("box 9.1 RGBA", 0, 0, 0, 448, 336), ("box 20 RGB8", 8, 8, 0, 205, 154),
    ("gamma 4:1", 0, 0, 1, 1024, 768), ("gamma ratio 1.5", 0, 0, 1, 2731, 2048), ("gamma ratio 2.5 RGB8", 8, 8, 1, 1638, 1229),
    ("gamma box 9.1", 0, 0, 1, 448, 336), ("gamma box 20 RGB8", 8, 8, 1, 205, 154), ("gamma x2", 0, 0, 1, 8192, 6144),
    ("premul16 ratio 1.5", 4, 4, 0, 2731, 2048), ("premul16 x2", 4, 4, 0, 8192, 6144), ("H down 1.5, V up 1.5 gamma", 0, 0, 1, 2731, 4608),
]
for (label, t_in, t_out, srgb, w_out, h_out) in CASES:
    row = {"case": label}
    for frames in (1, 8):
        cfg = (t_in, 4096, 3072, None, t_out, w_out, h_out, frames, "x") + ((srgb,) if srgb else ())
        B.CONFIGS["T"] = cfg
        line = B.run_config("T", 20, 3, settle=0.1, no_arena=True)
        row["us_%d" % frames] = line["step_us"]
        row["kernel"] = line["kernel"]
    row["single_over_eighth"] = round(row["us_1"] / (row["us_8"] / 8.0), 2)
    print(json.dumps(row), flush=True)
