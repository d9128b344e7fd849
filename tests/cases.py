"""Shared case lists + restatements of the reference's own verification inputs."""
import numpy as np

import helpers as H

GOLDEN = H.os.path.join(H.ROOT, "tests", "golden")


def load_golden(name):
    return np.load(H.os.path.join(GOLDEN, name))


def matrix_cases(stride=1):
    g = load_golden("matrix.npz")
    cases, offs, blob = g["cases"], g["offsets"], g["blob"]
    for i in range(0, len(cases), stride):
        t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, valid = [int(v) for v in cases[i]]
        yield (t_in, w_in, h_in, t_out, w_out, h_out, srgb, seed, bool(valid)), blob[offs[i]:offs[i + 1]]


# verify.c:57-98 populate_pixels()
_POS = {0: (0, 1, 2, 3), 1: (2, 1, 0, 3), 2: (1, 2, 3, 0), 3: (3, 2, 1, 0)}


def ordering_pattern(t, n_bytes_max=65536):
    bpp = H.bpp(t)
    n_pixels = n_bytes_max // bpp
    buf = np.zeros(n_bytes_max + 16, np.uint8)
    m = (np.arange(n_pixels) % 16) * 4
    v = buf[:n_pixels * bpp].reshape(-1, bpp)
    if bpp == 4:
        r, g, b, a = _POS[t % 4]
        v[:, a] = 0xff
    else:
        r, g, b = (0, 1, 2) if t == H.RGB8 else (2, 1, 0)
    v[:, r] = 0x20 + m
    v[:, g] = 0x60 + m
    v[:, b] = 0xa0 + m
    return buf


def unassoc_alpha_cases():
    """verify.c:227-301: (input 8 bytes, expected 4 bytes, tolerance)"""
    for i in range(256):
        exp0 = i // 2
        col = 0 if exp0 == 0 else 0xff
        tol = 0x7f if i < 0x0a else 0x16 if i < 0x20 else 0x10 if i < 0x30 else 0x08 if i < 0x40 else 4
        yield [i, 0xff, 0xff, 0xff, 0, 0, 0, 0], [exp0, col, col, col], tol
    for i in range(256):
        c = (0xff * 0xff) // (0xff + i)
        yield [0xff, 0xff, 0xff, 0xff, i, 0, 0, 0], [(0xff + i) // 2, c, c, c], 1


# verify.c:343-395 / 463-514 size set
VERIFY_SIZES = ((1, 65535), (2, 65535), (65534, 65535), (65535, 1), (65535, 65534))
