#!/usr/bin/env python3
"""Device-resident timing of every BASELINE.json config (+ the BOX row) on ONE GPU.

Not the driver's benchmark (that is bench.py, config 2 only): this is the tool the
per-config numbers in DESIGN.md / profiles/ come from. One JSON line per config:
kernel class, launches, average launch time (hipEvents on the launch stream),
algorithmic GB/s and fraction of the 8 TB/s HBM peak, Mpix_out/s.

    python tools/bench_configs.py [--configs C1,C2,C3,C4,C5,BOX] [--iters 20]

This tool only measures. `python tests/bench_configs_checked.py <same arguments>` runs the same
configs and also compares one image per config with the oracle (the oracle is test
infrastructure and is only ever loaded from tests/).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import smolscale_amd as sm  # noqa: E402

HBM_PEAK_GBS = 8000.0

# name: (t_in, w_in, h_in, stride_in, t_out, w_out, h_out, n_images, description)
CONFIGS = {
    "C1": (0, 640, 480, None, 0, 320, 240, 384, "640x480->320x240 RGBA8 premul (0H/0H), 384 frames per launch"),
    "C2": (0, 4096, 4096, None, 0, 1024, 1024, 8, "4096^2->1024^2 RGBA8 premul (1H/1H), 8 frames per launch"),
    "C3": (4, 1920, 1080, None, 0, 7680, 4320, 4, "1920x1080 RGBA unassociated -> 7680x4320 RGBA premul (0H/0H magnify), 4 frames"),
    "C4": (8, 65535, 4096, 65535 * 3, 0, 8192, 4096, 1, "65535x4096 RGB8 (odd stride 196605) -> 8192x4096 RGBA8 (2H/COPY), 1 frame"),
    # a sub-rectangle of a larger image (pixels_in + 4 * x with the parent's stride) and rows whose stride is not a
    # multiple of 16: the alignment the streaming kernels used to demand (11th field: byte offset of the source base)
    "C2CROP4": (0, 4096, 4096, 4112 * 4, 0, 1024, 1024, 8, "4096^2->1024^2 RGBA8 premul, crop of a 4112-wide parent at x=1 (base + 4 B), 8 frames", 0, 4),
    "C2CROP12": (0, 4096, 4096, 4112 * 4, 0, 1024, 1024, 8, "4096^2->1024^2 RGBA8 premul, crop of a 4112-wide parent at x=3 (base + 12 B), 8 frames", 0, 12),
    "C2CROP0": (0, 4096, 4096, 4112 * 4, 0, 1024, 1024, 8, "4096^2->1024^2 RGBA8 premul, crop of a 4112-wide parent at x=0 (aligned; control), 8 frames", 0, 0),
    "C2ODD": (0, 4096, 4096, 4097 * 4, 0, 1024, 1024, 8, "4096^2->1024^2 RGBA8 premul, stride 16388 (every row another 16-byte phase), 8 frames", 0, 0),
    # slight magnifications (the upper half of the reference's own proportional sweep, bench.sh:39-43: scales 1 .. 2)
    "U1.1": (0, 3840, 2160, None, 0, 4224, 2376, 6, "3840x2160->4224x2376 RGBA8 premul (magnify x1.1), 6 frames"),
    "U1.3": (0, 3840, 2160, None, 0, 4992, 2808, 6, "3840x2160->4992x2808 RGBA8 premul (magnify x1.3), 6 frames"),
    "U1.5": (0, 3840, 2160, None, 0, 5760, 3240, 6, "3840x2160->5760x3240 RGBA8 premul (magnify x1.5), 6 frames"),
    "U1.75": (0, 3840, 2160, None, 0, 6720, 3780, 6, "3840x2160->6720x3780 RGBA8 premul (magnify x1.75), 6 frames"),
    "U1.37": (0, 3840, 2160, None, 0, 5261, 2959, 6, "3840x2160->5261x2959 RGBA8 premul (magnify x1.37, output rows not 16-byte aligned), 6 frames"),
    "U2.6": (0, 1920, 1080, None, 0, 4993, 2809, 8, "1920x1080->4993x2809 RGBA8 premul (magnify x2.6, output rows not 16-byte aligned), 8 frames"),
    # points of the reference's proportional sweep (ARGB8 premultiplied, test.c:20) that read low in profiles/r03*_sweep
    "S8K93": (2, 7680, 4320, None, 2, 7145, 4019, 5, "sweep point 7680x4320->7145x4019 ARGB8 premul (ratio 1.075), 5 frames"),
    "S4K97": (2, 3840, 2160, None, 2, 3724, 2094, 19, "sweep point 3840x2160->3724x2094 ARGB8 premul (ratio 1.03), 19 frames"),
    "S4K89": (2, 3840, 2160, None, 2, 3417, 1922, 19, "sweep point 3840x2160->3417x1922 ARGB8 premul (ratio 1.12), 19 frames"),
    "S4K81": (2, 3840, 2160, None, 2, 3110, 1749, 19, "sweep point 3840x2160->3110x1749 ARGB8 premul (ratio 1.235), 19 frames"),
    "S4K73": (2, 3840, 2160, None, 2, 2803, 1577, 19, "sweep point 3840x2160->2803x1577 ARGB8 premul (ratio 1.37), 19 frames"),
    "S4K65": (2, 3840, 2160, None, 2, 2496, 1404, 19, "sweep point 3840x2160->2496x1404 ARGB8 premul (ratio 1.54), 19 frames"),
    "S4K57": (2, 3840, 2160, None, 2, 2188, 1231, 19, "sweep point 3840x2160->2188x1231 ARGB8 premul (ratio 1.755), 19 frames"),
    "S4K110": (2, 3840, 2160, None, 2, 4224, 2376, 12, "sweep point 3840x2160->4224x2376 ARGB8 premul (magnify x1.1), 12 frames"),
    "S4K109ODD": (2, 3840, 2160, None, 2, 4185, 2354, 12, "sweep point 3840x2160->4185x2354 ARGB8 premul (magnify x1.09, odd width: rows start on 4-byte boundaries only), 12 frames"),
    "S4K91ODD": (2, 3840, 2160, None, 2, 3494, 1965, 12, "sweep point 3840x2160->3494x1965 ARGB8 premul (ratio 1.099, rows of 13976 bytes), 12 frames"),
    "S4K140": (2, 3840, 2160, None, 2, 5376, 3024, 12, "sweep point 3840x2160->5376x3024 ARGB8 premul (magnify x1.4), 12 frames"),
    "S4K176": (2, 3840, 2160, None, 2, 6758, 3801, 19, "sweep point 3840x2160->6758x3801 ARGB8 premul (magnify x1.76), 19 frames"),
    "S4K193": (2, 3840, 2160, None, 2, 7411, 4168, 19, "sweep point 3840x2160->7411x4168 ARGB8 premul (magnify x1.93), 19 frames"),
    "S1080T": (2, 1920, 1080, None, 2, 76, 42, 64, "sweep point 1920x1080->76x42 ARGB8 premul (box ratio 25), 64 frames"),
    "S4K49": (2, 3840, 2160, None, 2, 1881, 1057, 19, "sweep point 3840x2160->1881x1057 ARGB8 premul (ratio 2.04, 1H), 19 frames"),
    "S4K25": (2, 3840, 2160, None, 2, 959, 539, 19, "sweep point 3840x2160->959x539 ARGB8 premul (ratio 4.004, 2H), 19 frames"),
    "U2": (0, 3840, 2160, None, 0, 7680, 4320, 6, "3840x2160->7680x4320 RGBA8 premul (magnify x2), 6 frames"),
    "U4RGB": (8, 1920, 1080, None, 8, 7680, 4320, 4, "1920x1080->7680x4320 RGB8 -> RGB8 (magnify x4, 24-bit rows), 4 frames"),
    "U2UNASSOC": (0, 3840, 2160, None, 4, 7680, 4320, 6, "3840x2160->7680x4320 RGBA8 premul -> unassociated (magnify x2), 6 frames"),
    "U8": (0, 960, 540, None, 0, 7680, 4320, 4, "960x540->7680x4320 RGBA8 premul (magnify x8), 4 frames"),
    "U3": (0, 1920, 1080, None, 0, 5760, 3240, 8, "1920x1080->5760x3240 RGBA8 premul (magnify x3), 8 frames"),
    "C5": (0, 256, 256, None, 0, 64, 64, 10000, "10000 thumbnails 256^2->64^2 RGBA8 premul (1H/1H), one batch"),
    "C5mix": (-1, 256, 256, None, -1, 64, 64, 10000, "10000 thumbnails 256^2->64^2, pixel types drawn uniformly from the 10x10 table"),
    "BOX": (0, 4096, 4096, None, 0, 448, 448, 8, "4096^2->448^2 RGBA8 premul (BOX/BOX 64bpp), 8 frames per launch"),
    # arbitrary (non power-of-two) ratios through k_tile
    "R1.5": (0, 4096, 4096, None, 0, 2731, 2731, 6, "4096^2->2731^2 RGBA8 premul (0H, ratio 1.5), 6 frames"),
    "R1.03": (0, 4096, 4096, None, 0, 3977, 3977, 4, "4096^2->3977^2 RGBA8 premul (0H, ratio 1.03), 4 frames"),
    "R1.1": (0, 4096, 4096, None, 0, 3724, 3724, 4, "4096^2->3724^2 RGBA8 premul (0H, ratio 1.1), 4 frames"),
    "R1.2": (0, 4096, 4096, None, 0, 3413, 3413, 6, "4096^2->3413^2 RGBA8 premul (0H, ratio 1.2), 6 frames"),
    "R1.9": (0, 4096, 4096, None, 0, 2156, 2156, 6, "4096^2->2156^2 RGBA8 premul (0H, ratio 1.9), 6 frames"),
    "R1.5RGB": (8, 4096, 4096, None, 0, 2731, 2731, 6, "4096^2 RGB8 ->2731^2 RGBA8 (0H, ratio 1.5), 6 frames"),
    "R1.5U": (4, 4096, 4096, None, 0, 2731, 2731, 6, "4096^2 RGBA unassociated ->2731^2 RGBA8 premul (0H, ratio 1.5), 6 frames"),
    "HDOWNVUP": (0, 4096, 2048, None, 0, 2731, 3000, 6, "4096x2048->2731x3000 RGBA8 premul (H ratio 1.5 down, V up), 6 frames"),
    "R3": (0, 4096, 4096, None, 0, 1365, 1365, 8, "4096^2->1365^2 RGBA8 premul (1H, ratio 3), 8 frames"),
    "R1.35": (0, 4096, 4096, None, 0, 3034, 3034, 6, "4096^2->3034^2 RGBA8 premul (0H, ratio 1.35), 6 frames"),
    "R1.7": (0, 4096, 4096, None, 0, 2409, 2409, 6, "4096^2->2409^2 RGBA8 premul (0H, ratio 1.7), 6 frames"),
    "R2.5": (0, 4096, 4096, None, 0, 1638, 1638, 8, "4096^2->1638^2 RGBA8 premul (1H, ratio 2.5), 8 frames"),
    "R3.5": (0, 4096, 4096, None, 0, 1170, 1170, 8, "4096^2->1170^2 RGBA8 premul (1H, ratio 3.5), 8 frames"),
    "R5": (0, 4096, 4096, None, 0, 819, 819, 8, "4096^2->819^2 RGBA8 premul (2H, ratio 5), 8 frames"),
    "R7": (0, 4096, 4096, None, 0, 585, 585, 8, "4096^2->585^2 RGBA8 premul (2H, ratio 7), 8 frames"),
    "R6": (0, 4096, 4096, None, 0, 683, 683, 8, "4096^2->683^2 RGBA8 premul (2H, ratio 6), 8 frames"),
    "PHOTO": (8, 4000, 3000, None, 8, 1600, 1200, 12, "4000x3000 RGB8 -> 1600x1200 RGB8 (1H, ratio 2.5), 12 frames"),
    "PHOTOODD": (8, 4000, 3000, None, 8, 1601, 1201, 12, "4000x3000 RGB8 -> 1601x1201 RGB8 (1H; 24-bit output rows that do not start on 4-byte boundaries), 12 frames"),
    "PHOTO32ODD": (8, 4001, 3001, 4001 * 3, 0, 1601, 1201, 12, "4001x3001 RGB8 (stride 12003) -> 1601x1201 RGBA8 (1H; odd 24-bit input rows), 12 frames"),
    "PNG": (4, 4000, 3000, None, 0, 1600, 1200, 10, "4000x3000 RGBA unassociated -> 1600x1200 RGBA premul (1H), 10 frames"),
    # 128bpp-class rows: premul16 and sRGB-linear internal formats
    "P16": (4, 4096, 4096, None, 4, 1024, 1024, 4, "4096^2->1024^2 RGBA8 unassociated->unassociated (premul16, 128bpp class), 4 frames"),
    "SRGB": (0, 4096, 4096, None, 0, 1024, 1024, 4, "4096^2->1024^2 RGBA8 premul with_srgb=1 (linear, 128bpp class), 4 frames", 1),
    "P16UP": (4, 1920, 1080, None, 4, 7680, 4320, 2, "1920x1080->7680x4320 RGBA8 unassociated->unassociated (premul16 magnify), 2 frames"),
    "SRGBUP": (0, 1920, 1080, None, 0, 7680, 4320, 2, "1920x1080->7680x4320 RGBA8 premul with_srgb=1 (linear magnify), 2 frames", 1),
    "SRGBR6": (0, 4096, 4096, None, 0, 683, 683, 4, "4096^2->683^2 RGBA8 premul with_srgb=1 (gamma-correct, 2 halvings, ratio 6), 4 frames", 1),
    "SRGBR7RGB": (8, 4000, 3000, None, 8, 560, 420, 8, "4000x3000->560x420 RGB8 with_srgb=1 (gamma-correct photo, ratio 7.1), 8 frames", 1),
    "SRGBU1.1": (0, 3840, 2160, None, 0, 4224, 2376, 6, "3840x2160->4224x2376 RGBA8 premul with_srgb=1 (gamma-correct magnify x1.1), 6 frames", 1),
    "SRGBU2": (0, 3840, 2160, None, 0, 7680, 4320, 6, "3840x2160->7680x4320 RGBA8 premul with_srgb=1 (gamma-correct magnify x2), 6 frames", 1),
    "SRGBUP24": (8, 1920, 1080, None, 8, 7680, 4320, 4, "1920x1080->7680x4320 RGB8 with_srgb=1 (gamma-correct magnify x4, 24-bit rows), 4 frames", 1),
    "SRGBUPODD": (0, 1920, 1080, None, 0, 4993, 2809, 4, "1920x1080->4993x2809 RGBA8 premul with_srgb=1 (linear magnify x2.6, output rows start on 4-byte boundaries only), 4 frames", 1),
    "P16UPODD": (4, 1920, 1080, None, 4, 4993, 2809, 4, "1920x1080->4993x2809 rgbA->rgbA (premul16 magnify x2.6, odd width), 4 frames"),
    "P16HALFODD": (4, 4096, 4096, None, 4, 2733, 2733, 2, "4096^2->2733^2 unassociated->unassociated (premul16, ratio 1.5, odd width), 2 frames"),
    "P16HALF": (4, 4096, 4096, None, 4, 2731, 2731, 2, "4096^2->2731^2 unassociated->unassociated (premul16, ratio 1.5), 2 frames"),
    "SRGBPNGUP": (4, 1920, 1080, None, 4, 7680, 4320, 2, "1920x1080->7680x4320 rgbA->rgbA with_srgb=1 (premul16 + linear magnify), 2 frames", 1),
    "THUMB": (8, 4000, 3000, None, 8, 200, 150, 16, "4000x3000 RGB8 -> 200x150 RGB8 (BOX ratio 20, 64bpp class), 16 frames"),
    "SRGBTHUMB": (8, 4000, 3000, None, 8, 200, 150, 16, "4000x3000 RGB8 -> 200x150 RGB8 with_srgb=1 (BOX ratio 20, 128bpp class: gamma-correct photo thumbnails), 16 frames", 1),
    "SRGBBOX": (0, 4096, 4096, None, 0, 448, 448, 8, "4096^2->448^2 RGBA8 premul with_srgb=1 (BOX/BOX 128bpp), 8 frames", 1),
    "SRGBHALF": (0, 4096, 4096, None, 0, 2731, 2731, 2, "4096^2->2731^2 RGBA8 premul with_srgb=1 (linear, ratio 1.5), 2 frames", 1),
    "SRGBPHOTO": (8, 4000, 3000, None, 8, 1600, 1200, 4, "4000x3000 RGB8 -> 1600x1200 RGB8 with_srgb=1 (gamma-correct photo downscale), 4 frames", 1),
    "SRGBPNG": (4, 4000, 3000, None, 4, 1600, 1200, 4, "4000x3000 rgbA -> 1600x1200 rgbA with_srgb=1 (premul16 + linear), 4 frames", 1),
    # the correctness backstops (k_wtile / k_general<true>): 128bpp-class jobs whose axes differ -- no proportional resize lands here
    # (profiles/r04_kernel_histogram.txt), independently scaled axes do
    "SRGBHDVU": (0, 4096, 2048, None, 0, 2731, 3000, 2, "4096x2048->2731x3000 RGBA8 premul with_srgb=1 (H ratio 1.5 down, V up), 2 frames", 1),
    "SRGBVCOPY": (0, 4096, 4096, None, 0, 2731, 4096, 2, "4096^2->2731x4096 RGBA8 premul with_srgb=1 (H ratio 1.5, V COPY), 2 frames", 1),
    "SRGBBOXBIL": (0, 4096, 4096, None, 0, 300, 2000, 2, "4096^2->300x2000 RGBA8 premul with_srgb=1 (H BOX ratio 13.7, V bilinear), 2 frames", 1),
    "P16BOXBIL": (4, 4096, 4096, None, 4, 300, 2000, 2, "4096^2->300x2000 rgbA->rgbA (premul16; H BOX, V bilinear), 2 frames"),
    "P16HDVU": (4, 4096, 2048, None, 4, 2731, 3000, 2, "4096x2048->2731x3000 rgbA->rgbA (premul16; H ratio 1.5 down, V up), 2 frames"),
    "FLATBOXBIL": (0, 65535, 64, None, 0, 128, 40, 8, "65535x64->128x40 RGBA8 premul (H BOX ratio 512: the 128bpp class's compressed storage; V bilinear), 8 frames"),
    "SRGBBILBOX": (0, 4096, 4096, None, 0, 2731, 300, 2, "4096^2->2731x300 RGBA8 premul with_srgb=1 (H ratio 1.5, V BOX ratio 13.7), 2 frames", 1),
    "P16BILBOX": (4, 4096, 4096, None, 4, 2731, 300, 2, "4096^2->2731x300 rgbA->rgbA (premul16; H ratio 1.5, V BOX ratio 13.7), 2 frames"),
    "P16LBILBOX": (4, 4096, 4096, None, 4, 2731, 300, 2, "4096^2->2731x300 rgbA->rgbA with_srgb=1 (premul16 + gamma: 19-bit channels; H ratio 1.5, V BOX ratio 13.7), 2 frames", 1),
    "FLATBILBOX": (0, 300, 20000, None, 0, 200, 40, 8, "300x20000->200x40 RGBA8 premul (H ratio 1.5, V BOX ratio 500: compressed storage), 8 frames"),
    "FLATTALL": (0, 64, 60000, None, 0, 100, 4, 16, "64x60000->100x4 RGBA8 premul (H magnify, V BOX ratio 15000: compressed storage, few output pixels), 16 frames"),
    "BOX12": (0, 4096, 4096, None, 0, 336, 336, 8, "4096^2->336^2 RGBA8 premul (BOX ratio 12.2, 64bpp class: the two-pass threshold), 8 frames"),
    "BOX16": (0, 4096, 4096, None, 0, 256, 256, 8, "4096^2->256^2 RGBA8 premul (BOX ratio 16, 64bpp class), 8 frames"),
    "SRGBBOX64": (0, 4096, 4096, None, 0, 64, 64, 8, "4096^2->64^2 RGBA8 premul with_srgb=1 (BOX ratio 64, 128bpp class), 8 frames", 1),
    "SRGBBOX128": (0, 8192, 8192, None, 0, 64, 64, 2, "8192^2->64^2 RGBA8 premul with_srgb=1 (BOX ratio 128, 128bpp class), 2 frames", 1),
    "WIDEBOX": (0, 65535, 512, None, 0, 64, 8, 4, "65535x512->64x8 RGBA8 premul (BOX ratio 1024 x 64, 128bpp class, 32 KB of source per output-column group and row), 4 frames"),
    "BOX128": (0, 8192, 8192, None, 0, 30, 30, 2, "8192^2->30^2 RGBA8 premul (BOX ratio 273 > 255: 128bpp class), 2 frames"),
}


def bpp(t):
    return sm.bytes_per_pixel(t)


def gpu_state():
    """sclk / socket power / junction temperature as rocm-smi reports them right after a timing: the boxes of the pool
    differ by 10-15 % on instruction-bound kernels (C5mix: 396-413 us on some, 456-471 on others, whatever the process,
    allocation or image order: profiles/r04_c5mix_modes.txt), and this is what tells them apart in a table."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        d = {}
        for l in out.splitlines():
            v = l.split(":", 2)[-1].strip()
            if "sclk" in l:
                d["sclk_mhz"] = int("".join(c for c in v.split("(")[-1] if c.isdigit()) or 0)
            elif "Power (W)" in l:
                d["power_w"] = float(v)
            elif "junction" in l.lower():
                d["temp_c"] = float(v)
        return d
    except Exception:       # noqa: BLE001
        return {}


def natural_bytes(n_bytes, bytes_pp, width, rng):
    """Image-like bytes instead of white noise: every channel a smooth ramp across the row and down the image plus a few
    grey levels of noise, alpha (32-bit pixels) mostly opaque with soft-edged translucent regions. Neighbouring pixels then
    hold neighbouring values, as in photographs -- which is what decides how the gamma-correct kernels' table look-ups
    spread over the LDS banks (white noise is their worst case)."""
    n_px = n_bytes // bytes_pp + 1
    i = np.arange(n_px, dtype=np.int64)
    x, y = i % max(1, width), i // max(1, width)
    out = np.empty((n_px, bytes_pp), np.uint8)
    for c in range(bytes_pp):
        ramp = (x * (3 + 2 * c)) // 16 + (y * (5 + c)) // 24 + 40 * c
        out[:, c] = ((ramp + rng.integers(-3, 4, n_px)) % 256).astype(np.uint8)
    if bytes_pp == 4:
        a = np.clip(255 - np.maximum(0, 96 - np.abs(((x + y) % 512) - 256)) * 2, 0, 255).astype(np.uint8)
        out[:, 3] = a
        out[:, 0] = a       # (either end may be the alpha byte: both carry the alpha pattern, colours in between)
    return out.reshape(-1)[:n_bytes]


IMAGES_OVERRIDE = None      # --images N


REGION_BYTES = 104 << 30
ROTATE = True
_REGION = []


def placement_region():
    """one 104 GB allocation per process (kept until exit): more than a third of the device, so its head and its tail lie in
    different zones of the device's memory wherever the driver puts it (profiles/r06_headline_modes.txt)"""
    if not _REGION:
        _REGION.append(sm.DeviceBuffer(REGION_BYTES))
    return _REGION[0]


def run_config(name, iters, warmup, checker=None, graph=False, tin=None, tout=None, srgb_override=None, settle=0.3, opaque=False, no_arena=False, natural=False, placement="off"):
    if name.startswith("G:"):
        # ad-hoc geometry: G:t_in:w_in:h_in:t_out:w_out:h_out:n_images[:srgb]
        f = [int(v) for v in name.split(":")[1:]]
        cfg = (f[0], f[1], f[2], None, f[3], f[4], f[5], f[6], "ad hoc %dx%d type %d -> %dx%d type %d, %d frames" % (f[1], f[2], f[0], f[4], f[5], f[3], f[6])) + ((f[7],) if len(f) > 7 else ())
    else:
        cfg = CONFIGS[name]
    t_in0, w_in, h_in, stride_in, t_out0, w_out, h_out, n_images, desc = cfg[:9]
    if IMAGES_OVERRIDE:
        n_images = IMAGES_OVERRIDE                  # launch-size sensitivity (tail effects of small launches)
    srgb = cfg[9] if len(cfg) > 9 else 0
    src_off = cfg[10] if len(cfg) > 10 else 0
    if tin is not None:
        t_in0, stride_in = tin, None
    if tout is not None:
        t_out0 = tout
    if srgb_override is not None:
        srgb = srgb_override
    if tin is not None or tout is not None or srgb_override is not None:
        desc += " [types overridden: in=%d out=%d srgb=%d]" % (t_in0, t_out0, srgb)
    rng = np.random.default_rng(12345)
    trng = np.random.default_rng(20260202)      # pixel types of the mixed batches: the draw bench.py's side table makes
    images, bufs, specs = [], [], []
    total_in = total_out = 0
    host_src_kept = None

    # Small images come out of one arena per side (256-byte aligned slots), as a pipeline that
    # batches thumbnails would hold them; big ones get their own allocations.
    class _Slot:
        def __init__(self, arena, off, n):
            self.arena, self.off, self.ptr, self.n = arena, off, arena.ptr + off, n
        def upload(self, host, offset=0):
            self.arena.upload(host, offset=self.off + offset)
        def download(self, n_bytes=None):
            return self.arena.download(self.n if n_bytes is None else n_bytes, offset=self.off)
        def free(self):
            pass
    arena_in = arena_out = None
    use_arena = (w_in * h_in * 4 <= (1 << 22)) and not no_arena
    # placement (round 6, profiles/r06_headline_modes.txt): source and destination memory of a config as windows of ONE large region --
    # "far": the sources at its head, the destinations at its end (another zone of the device's memory: the streaming kernels' fast
    # mode), "near": right behind the sources (the same zone: the slow mode); "off": separate allocations wherever the allocator puts them
    region = placement_region() if placement in ("far", "near") else None
    reg_in = reg_out = 0
    slot_in = ((h_in * (stride_in or w_in * 4) + 64 + 255) // 256) * 256        # sized for 4-byte pixels
    slot_out = ((h_out * w_out * 4 + 64 + 255) // 256) * 256
    if os.environ.get("SMOL_BENCH_SLOT_PAD"):           # experiment: another pitch between the images of an arena
        slot_in += int(os.environ["SMOL_BENCH_SLOT_PAD"]) // 256 * 256
        slot_out += int(os.environ.get("SMOL_BENCH_SLOT_PAD_OUT", "0")) // 256 * 256
    if region is not None:
        in_total = n_images * (slot_in + 256)
        out_total = n_images * (slot_out + 256)
        reg_out = ((region.n_bytes - out_total - (1 << 21)) if placement == "far" else (in_total + (1 << 21))) >> 21 << 21
        if in_total + out_total + (8 << 20) > region.n_bytes:
            region = None               # (does not fit the region: separate allocations)
    if use_arena and region is None:
        arena_in, arena_out = sm.DeviceBuffer(n_images * slot_in), sm.DeviceBuffer(n_images * slot_out)
        if os.environ.get("SMOL_BENCH_PRINT_ARENA"):
            print("arena %s in %x out %x" % (name, arena_in.ptr, arena_out.ptr), file=sys.stderr)
    off_in = off_out = 0

    # one shared pool of random bytes uploaded once per distinct input size
    for i in range(n_images):
        t_in = int(trng.integers(0, 10)) if t_in0 < 0 else t_in0
        t_out = int(trng.integers(0, 10)) if t_out0 < 0 else t_out0
        s_in = stride_in or w_in * bpp(t_in)
        s_out = w_out * bpp(t_out)
        n_in = h_in * s_in
        n_out = h_out * s_out
        if region is not None:
            dsrc = _Slot(region, reg_in, n_in + 64 + src_off)
            ddst = _Slot(region, reg_out, n_out + 64)
            reg_in += (n_in + 64 + src_off + 255) // 256 * 256 if not use_arena else slot_in
            reg_out += (n_out + 64 + 255) // 256 * 256 if not use_arena else slot_out
        elif use_arena:
            dsrc = _Slot(arena_in, off_in, n_in)
            ddst = _Slot(arena_out, off_out, n_out)
            off_in += slot_in
            off_out += slot_out
        else:
            dsrc = sm.DeviceBuffer(n_in + 64 + src_off)
            ddst = sm.DeviceBuffer(n_out + 64)
        if i < 16 or n_in > (1 << 24):
            host = natural_bytes(n_in, bpp(t_in), w_in, rng) if natural else rng.integers(0, 256, n_in, dtype=np.uint8)
            pool = host
        else:
            host = pool          # reuse host bytes (device copies are still distinct allocations)
        if opaque and bpp(t_in) == 4 and (i < 16 or n_in > (1 << 24)):
            # photographs with an alpha channel are opaque: alpha byte 3 (RGBA/BGRA) or 0 (ARGB/ABGR)
            host[(3 if (t_in & 3) < 2 else 0)::4] = 255
        dsrc.upload(host[:n_in], offset=src_off) if src_off else dsrc.upload(host[:n_in])
        if i == 0:
            host_src_kept = host[:n_in].copy()
        im = sm.SmolHipImage()
        im.job = sm.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        im.src, im.dst = dsrc.ptr + src_off, ddst.ptr
        im.rowstride_in, im.rowstride_out = s_in, s_out
        im.first_row, im.n_rows = 0, h_out
        images.append(im)
        bufs.append((dsrc, ddst))
        specs.append((t_in, t_out, s_in, s_out))
        total_in += w_in * h_in * bpp(t_in)
        total_out += w_out * h_out * bpp(t_out)

    batch = sm.HipBatch(images)
    placement_us = None
    if region is not None and placement == "far":
        # "far" = the best of six destination positions spread over the region (the zones of the device's memory do not lie where
        # one would guess: bench.py's pick_placement scans the same way). Only the destinations move; the probes are a dozen launches
        # after 0.03 s of load each.
        lo, hi = (in_total + (1 << 21)) >> 21 << 21, (region.n_bytes - out_total - (1 << 21)) >> 21 << 21
        bases = [hi] + [(lo + (hi - lo) * k // 5) >> 21 << 21 for k in range(5)]
        rel = [b.off - bufs[0][1].off for (_, b) in bufs]
        best, placement_us = None, []
        for base in bases:
            for im, r in zip(images, rel):
                im.dst = region.ptr + base + r
            cand = sm.HipBatch(images)
            p_ms = max(cand.time(1, 3) / 3, 1e-3)
            cand.time(0, max(1, int(30.0 / p_ms)))
            n = max(4, min(12, int(6.0 / p_ms)))
            t = cand.time(1, n) / n
            placement_us.append(round(t * 1e3, 1))
            cand.destroy()
            if best is None or t < best[0]:
                best = (t, base)
        batch.destroy()
        for k, (im, r) in enumerate(zip(images, rel)):
            im.dst = region.ptr + best[1] + r
            bufs[k] = (bufs[k][0], _Slot(region, best[1] + r, bufs[k][1].n))
        batch = sm.HipBatch(images)
    # The timed launches rotate over enough destination sets that 512 MB of other output lie between two writes of the same line: a
    # batch launched again and again into one small arena has its writes absorbed by the 256 MB Infinity Cache (round 6,
    # profiles/r06_dst_rotation.txt: the headline 324 -> 334 us, the sweep's cliff where 19 outputs pass 256 MB). --no-rotate = one set.
    dst0_ptrs = [im.dst for im in images]
    one_set = max(1, sum(getattr(b, "n", None) or b.n_bytes for (_, b) in bufs))
    n_rot = max(1, min(32, -(-(512 << 20) // one_set))) if ROTATE else 1
    extra, rot_batches = [], [batch]
    for k in range(1, n_rot):
        if region is not None:
            set_pitch = (out_total + (1 << 21)) >> 21 << 21
            base0 = bufs[0][1].off
            down = base0 - (n_rot - 1) * set_pitch >= in_total + (1 << 21)
            shift = (-k if down else k) * set_pitch
            if not down and base0 + (n_rot - 1) * set_pitch + out_total > region.n_bytes:
                break                               # (no room: fewer sets)
            for im, p0 in zip(images, dst0_ptrs):
                im.dst = p0 + shift
        elif arena_out is not None:
            a = sm.DeviceBuffer(n_images * slot_out)
            extra.append(a)
            for i, im in enumerate(images):
                im.dst = a.ptr + i * slot_out
        else:
            for im, (_, b) in zip(images, bufs):
                d = sm.DeviceBuffer(b.n_bytes)
                extra.append(d)
                im.dst = d.ptr
        rot_batches.append(sm.HipBatch(images))
    for im, p0 in zip(images, dst0_ptrs):
        im.dst = p0
    if len(rot_batches) > 1:
        batch = sm.HipBatchRotation(rot_batches)
    n_launches, kname = batch.info()
    if graph:
        batch.use_graph(True)
        kname += " (hipGraph: one kernel node per image)"
    # The GPU clock governor needs a few hundred ms of load to settle; instruction-bound
    # kernels measured right after start-up read 10-20 % slow. Warm up for `settle' seconds.
    probe_ms = batch.time(warmup, 5) / 5
    n_settle = int(settle * 1e3 / max(probe_ms, 1e-3))
    ms = batch.time(max(warmup, n_settle), iters)
    per_step_ms = ms / iters
    algo = total_in + total_out
    # SURVEY 8d asks for min and median too: 20 single-step timings (hipEvents around one step each)
    singles = sorted(x * 1e3 for x in batch.time_each(0, 20))
    state = {}
    if os.environ.get("SMOL_BENCH_GPU_STATE"):
        batch.time(0, max(20, iters))           # (the readout samples the clock the GPU runs at under this load)
        state = gpu_state()
    line = {
        "config": name, "desc": desc, "kernel": kname, "launches_per_step": n_launches,
        "step_us": round(per_step_ms * 1e3, 2),
        "step_us_min": round(singles[0], 2), "step_us_median": round(singles[len(singles) // 2], 2),
        "algorithmic_bytes_per_step": algo,
        "GBps": round(algo / (per_step_ms * 1e-3) / 1e9, 1),
        "frac_of_8TBps": round(algo / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "Mpix_out_per_s": round(n_images * w_out * h_out / (per_step_ms * 1e-3) / 1e6, 1),
        "images_per_s": round(n_images / (per_step_ms * 1e-3), 1),
    }
    if state:
        line["gpu_state"] = state
    if arena_in is not None and os.environ.get("SMOL_BENCH_ARENA_PROBE"):
        # what the address translation of THIS input arena costs (tools/microbench/tlb_probe.hip): ns per dependent read
        # at strides of 4 KiB, 64 KiB and 2 MiB (+ a cache line)
        import ctypes
        probe = ctypes.CDLL(os.path.abspath(os.environ["SMOL_BENCH_ARENA_PROBE"]))
        probe.smol_probe_latency.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
        res = {}
        for label, stride in (("4K", 4096 + 64), ("64K", 65536 + 64), ("2M", (2 << 20) + 64)):
            ns = ctypes.c_double(0.0)
            rc = probe.smol_probe_latency(arena_in.ptr, n_images * slot_in, stride, 20000, ctypes.byref(ns))
            res[label] = round(ns.value, 1) if rc == 0 else None
        line["arena_probe_ns"] = res
        line["arena_in_ptr"] = "%x" % arena_in.ptr
    if checker is not None:
        # checker (host_src, type_in, w_in, h_in, stride_in, type_out, w_out, h_out, stride_out, srgb, got) -> bool
        t_in, t_out, s_in, s_out = specs[0]
        got = bufs[0][1].download(h_out * s_out)
        line["bit_exact_first_rows_vs_oracle"] = bool(
            checker(host_src_kept, t_in, w_in, h_in, s_in, t_out, w_out, h_out, s_out, srgb, got))
    if region is not None:
        line["placement"] = placement
        if placement_us:
            line["placement_dst_windows_us"] = placement_us
    line["destination_sets"] = len(rot_batches)
    batch.destroy()
    for x in extra:
        x.free()
    for a, b in bufs:
        a.free()
        b.free()
    if arena_in is not None:
        arena_in.free()
        arena_out.free()
    return line


def main(argv=None, checker=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C3,C4,C5,C5mix,BOX")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="submit each batch as a hipGraph of per-image launches")
    ap.add_argument("--tin", type=int, default=None, help="override the input pixel type (0-9) of every config")
    ap.add_argument("--tout", type=int, default=None, help="override the output pixel type (0-9)")
    ap.add_argument("--srgb", type=int, default=None, help="override with_srgb")
    ap.add_argument("--no-arena", action="store_true", help="one allocation per image even for small images")
    ap.add_argument("--opaque", action="store_true", help="set the alpha channel of 32-bit inputs to 255 (like real photographs)")
    ap.add_argument("--natural", action="store_true", help="image-like source bytes (smooth ramps + a little noise) instead of white noise")
    ap.add_argument("--images", type=int, default=None, help="override the number of images per launch")
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of warm-up load before timing (clock settling)")
    ap.add_argument("--placement", default="off", choices=["off", "far", "near"],
                    help="sources and destinations as windows of ONE 104 GB region: 'far' = sources at its head, destinations at its end (the fast mode "
                         "of the streaming kernels), 'near' = destinations right behind the sources (the slow mode); default: separate allocations")
    ap.add_argument("--no-rotate", action="store_true", help="every launch into the same destinations (rounds 1-5; the Infinity Cache absorbs small outputs)")
    args = ap.parse_args(argv)
    global IMAGES_OVERRIDE, ROTATE
    ROTATE = not args.no_rotate
    IMAGES_OVERRIDE = args.images
    for name in args.configs.split(","):
        print(json.dumps(run_config(name, args.iters, args.warmup, checker, args.graph,
                                    args.tin, args.tout, args.srgb, args.settle, args.opaque, args.no_arena, args.natural, args.placement)), flush=True)


if __name__ == "__main__":
    main()
