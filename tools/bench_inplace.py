#!/usr/bin/env python3
"""Whole-image calls from host memory by job class: ms per image through smol_scale_batch (PCIe inclusive), for the A/B of the
mapped road (the kernel works on page-locked host rows across PCIe: the caller's own, or a ring slot for pageable memory) against
the DMA road (SMOL_HIP_MAPPED=0). One line per geometry.

    python tools/bench_inplace.py [--pageable]; SMOL_HIP_MAPPED=0 python tools/bench_inplace.py [--pageable]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_rows_per_call as B  # noqa: E402

GEOMS = {
    "c2 RGBA 4096^2 -> 1024^2 (k_pow2)": (0, 4096, 4096, 0, 1024, 1024),
    "1080p -> 720p RGBA (k_lind)": (0, 1920, 1080, 0, 1280, 720),
    "RGBA unassoc 2048^2 -> 1365^2 (k_lind wide)": (4, 2048, 2048, 4, 1365, 1365),
    "RGB8 2000x1500 -> 800x600 (24-bit both sides)": (8, 2000, 1500, 8, 800, 600),
    "RGB8 4000x3000 -> RGBA 1000x750": (8, 4000, 3000, 0, 1000, 750),
    "RGBA 1024^2 -> 2048^2 (k_up)": (0, 1024, 1024, 0, 2048, 2048),
    "RGBA 800x600 -> 2000x1500 (k_up, odd)": (0, 800, 600, 0, 2000, 1500),
    "RGBA 4000x3000 -> 400x300 (box)": (0, 4000, 3000, 0, 400, 300),
    "RGB8 4000x3000 -> RGB8 401x301 (box, 24-bit)": (8, 4000, 3000, 8, 401, 301),
    "RGBA 2048^2 -> 224^2 (box)": (0, 2048, 2048, 0, 224, 224),
    "RGBA 3840x2160 -> 1920x1080": (0, 3840, 2160, 0, 1920, 1080),
    "RGBA 640x480 -> 320x240": (0, 640, 480, 0, 320, 240),
    "RGB8 65535x512 -> RGBA 8192x512 (k_hspan)": (8, 65535, 512, 0, 8192, 512),
}

if __name__ == "__main__":
    pinned = "--pageable" not in sys.argv
    print("# %s memory, SMOL_HIP_MAPPED=%s SMOL_HIP_MAPPED_CHUNK_KB=%s" % ("page-locked" if pinned else "pageable", os.environ.get("SMOL_HIP_MAPPED", "(default)"),
                                                                  os.environ.get("SMOL_HIP_MAPPED_CHUNK_KB", "(default)")))
    for name, g in GEOMS.items():
        r = B.measure(g, pinned, 0, reps=5)
        mb = (g[1] * g[2] * (3 if g[0] >= 8 else 4) + g[4] * g[5] * (3 if g[3] >= 8 else 4)) / 1e6
        print("%-52s %8.3f ms  %6.1f GB/s over the link  ok=%s" % (name, r["ms_per_image"], mb / r["ms_per_image"], r["equal_to_whole_image_call"]), flush=True)
