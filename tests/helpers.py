"""Shared test plumbing: ctypes bindings for the oracle (oracle/libsmol_oracle.so),
the real reference when its prebuilt shared objects are present (oracle/_ref/), and
the product library (smolscale_amd/libsmolscale_amd.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
anything under oracle/ -- this module is the single place tests do so.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libsmol_oracle.so")
REF_GENERIC_SO = os.path.join(ROOT, "oracle", "_ref", "libsmolref_generic.so")
REF_AVX2_SO = os.path.join(ROOT, "oracle", "_ref", "libsmolref_avx2.so")

# smolscale.h:16-36
RGBA8_P, BGRA8_P, ARGB8_P, ABGR8_P = 0, 1, 2, 3
RGBA8_U, BGRA8_U, ARGB8_U, ABGR8_U = 4, 5, 6, 7
RGB8, BGR8 = 8, 9
N_TYPES = 10
TYPE_NAMES = ["RGBA8p", "BGRA8p", "ARGB8p", "ABGR8p", "RGBA8u", "BGRA8u", "ARGB8u", "ABGR8u", "RGB8", "BGR8"]

SEED0 = 0x9E3779B97F4A7C15


def bpp(t):
    return 3 if t >= 8 else 4


class OraclePlan(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "filter_h", "filter_v", "storage_type", "gamma_type",
        "width_bilin_out", "height_bilin_out", "width_halvings", "height_halvings",
        "span_mul_x", "span_mul_y", "premul16", "quirk24")]


class RefProbePlan(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "filter_h", "filter_v", "storage_type", "gamma_type",
        "width_bilin_out", "height_bilin_out", "width_halvings", "height_halvings",
        "span_mul_x", "span_mul_y")]


_u32 = ctypes.c_uint32
_vp = ctypes.c_void_p


def _build_oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            _build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.smol_oracle_plan.argtypes = [_u32] * 7 + [ctypes.POINTER(OraclePlan)]
        lib.smol_oracle_plan.restype = ctypes.c_int
        lib.smol_oracle_precalc.argtypes = [_u32] * 5 + [_vp, _u32]
        lib.smol_oracle_precalc.restype = _u32
        lib.smol_oracle_scale_rows.argtypes = [_vp, _u32, _u32, _u32, _u32,
                                               _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32]
        lib.smol_oracle_scale_rows.restype = ctypes.c_int
        lib.smol_oracle_fill.argtypes = [_vp, ctypes.c_uint64, ctypes.c_uint64]
        lib.smol_oracle_fill.restype = None
        lib.smol_oracle_make_valid_premul.argtypes = [_vp, _u32, _u32, _u32, _u32]
        lib.smol_oracle_make_valid_premul.restype = None
        lib.smol_oracle_fnv1a.argtypes = [_vp, _u32, _u32, _u32]
        lib.smol_oracle_fnv1a.restype = ctypes.c_uint64
        _oracle = lib
    return _oracle


_refs = {}


def have_ref(kind="generic"):
    return os.path.exists(REF_GENERIC_SO if kind == "generic" else REF_AVX2_SO)


def ref(kind="generic"):
    """The real reference (smolscale.h API) from oracle/_ref."""
    if kind not in _refs:
        lib = ctypes.CDLL(REF_GENERIC_SO if kind == "generic" else REF_AVX2_SO)
        lib.smol_scale_simple.argtypes = [_vp, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, ctypes.c_uint8]
        lib.smol_scale_simple.restype = None
        lib.smol_scale_new.argtypes = lib.smol_scale_simple.argtypes
        lib.smol_scale_new.restype = _vp
        lib.smol_scale_destroy.argtypes = [_vp]
        lib.smol_scale_destroy.restype = None
        lib.smol_scale_batch.argtypes = [_vp, _u32, _u32]
        lib.smol_scale_batch.restype = None
        lib.smol_scale_batch_full.argtypes = [_vp, _vp, _u32, _u32]
        lib.smol_scale_batch_full.restype = None
        lib.ref_probe_plan.argtypes = [_u32] * 7 + [ctypes.POINTER(RefProbePlan)]
        lib.ref_probe_plan.restype = None
        lib.ref_probe_precalc.argtypes = [_u32] * 5 + [_vp, _u32]
        lib.ref_probe_precalc.restype = _u32
        _refs[kind] = lib
    return _refs[kind]


def make_input(t, w, h, seed=0, valid=True, stride=None, pad=0):
    """Seeded synthetic frame as a flat uint8 array of h*stride (+pad) bytes."""
    if stride is None:
        stride = w * bpp(t)
    buf = np.zeros(h * stride + pad, dtype=np.uint8)
    oracle().smol_oracle_fill(buf.ctypes.data, buf.size, (SEED0 + seed) & 0xFFFFFFFFFFFFFFFF)
    if valid:
        oracle().smol_oracle_make_valid_premul(buf.ctypes.data, t, w, h, stride)
    return buf


def oracle_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0,
                 stride_in=None, stride_out=None, first=0, n=None, fill=0xA5):
    stride_in = stride_in or w_in * bpp(t_in)
    stride_out = stride_out or w_out * bpp(t_out)
    n = h_out - first if n is None else n
    dst = np.full(n * stride_out, fill, dtype=np.uint8)
    rc = oracle().smol_oracle_scale_rows(src.ctypes.data, t_in, w_in, h_in, stride_in,
                                         dst.ctypes.data, t_out, w_out, h_out, stride_out,
                                         srgb, first, n)
    assert rc == 0
    return dst


def ref_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0,
              stride_in=None, stride_out=None, first=0, n=None, fill=0xA5, kind="generic"):
    stride_in = stride_in or w_in * bpp(t_in)
    stride_out = stride_out or w_out * bpp(t_out)
    n = h_out - first if n is None else n
    # The reference's packers store whole uint32 words for 32bpp and may read rows
    # through uint32 loads: keep a little slack after both buffers.
    dst = np.full(n * stride_out + 16, fill, dtype=np.uint8)
    lib = ref(kind)
    ctx = lib.smol_scale_new(src.ctypes.data, t_in, w_in, h_in, stride_in,
                             None, t_out, w_out, h_out, stride_out, srgb)
    lib.smol_scale_batch_full(ctx, dst.ctypes.data, first, n)
    lib.smol_scale_destroy(ctx)
    return dst[:n * stride_out]


def fnv1a(buf, row_bytes, height, stride):
    return int(oracle().smol_oracle_fnv1a(buf.ctypes.data, row_bytes, height, stride))


# --------------------------------------------------------------------------- #
# Product library (C-ABI of include/smol_hip.h + include/smolscale.h) through   #
# the package's own ctypes binding                                              #
# --------------------------------------------------------------------------- #

import sys

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import smolscale_amd as sm  # noqa: E402
from smolscale_amd import SmolHipJob, SmolHipImage  # noqa: E402,F401

EMU_SO = os.path.join(ROOT, "tests", "emu", "libsmol_emu.so")

FILTER_COPY, FILTER_ONE, FILTER_BILINEAR, FILTER_BOX = 0, 1, 2, 3


def product():
    return sm.lib()


def plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb=0):
    return sm.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)


def hip_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0,
              stride_in=None, stride_out=None, first=0, n=None, fill=0xA5):
    """HIP path through the public smol_scale_new()/smol_scale_batch_full() C API
    (host pointers in, host pointers out)."""
    stride_in = stride_in or w_in * bpp(t_in)
    stride_out = stride_out or w_out * bpp(t_out)
    n = h_out - first if n is None else n
    dst = np.full(n * stride_out, fill, dtype=np.uint8)
    ctx = sm.SmolScaleCtx(src, t_in, w_in, h_in, stride_in, None, t_out, w_out, h_out, stride_out, srgb)
    ctx.batch_full(dst, first, n)
    ctx.destroy()
    return dst


def hip_scale_device(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0,
                     stride_in=None, stride_out=None, first=0, n=None, fill=0xA5,
                     src_misalign=0, dst_misalign=0, device=0):
    """HIP path through smol_hip_scale_rows_device(): caller-owned device buffers with
    the caller's own strides/alignment (what a zero-copy pipeline would hand over)."""
    import ctypes as C
    stride_in = stride_in or w_in * bpp(t_in)
    stride_out = stride_out or w_out * bpp(t_out)
    n = h_out - first if n is None else n
    lib = sm.lib()
    dsrc = sm.DeviceBuffer(src.size + 64, device)
    ddst = sm.DeviceBuffer(n * stride_out + 64, device)
    ddst.fill(fill)
    dsrc.upload(src, offset=src_misalign)
    job = plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    plan = C.c_void_p()
    sm.check(lib.smol_hip_plan_create(C.byref(job), device, C.byref(plan)), "plan_create")
    sm.check(lib.smol_hip_scale_rows_device(plan, dsrc.ptr + src_misalign, stride_in,
                                            ddst.ptr + dst_misalign, stride_out, first, n, None),
             "scale_rows_device")
    sm.check(lib.smol_hip_device_synchronize(device), "sync")
    out = ddst.download(n * stride_out, offset=dst_misalign)
    lib.smol_hip_plan_destroy(plan)
    dsrc.free()
    ddst.free()
    return out


_emu = None


def emu():
    """Host emulation of the kernel bodies (tests/emu) -- logic debugging without a GPU."""
    global _emu
    if _emu is None:
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
        lib = ctypes.CDLL(EMU_SO)
        lib.emu_scale_rows.argtypes = [ctypes.POINTER(SmolHipJob), _vp, _u32, _vp, _u32, _u32, _u32,
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        lib.emu_scale_rows.restype = ctypes.c_int
        _emu = lib
    return _emu


def emu_scale(src, t_in, w_in, h_in, t_out, w_out, h_out, srgb=0,
              stride_in=None, stride_out=None, first=0, n=None, fill=0xA5, force_general=0):
    stride_in = stride_in or w_in * bpp(t_in)
    stride_out = stride_out or w_out * bpp(t_out)
    n = h_out - first if n is None else n
    dst = np.full(n * stride_out + 16, fill, dtype=np.uint8)
    job = plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    k = ctypes.c_int(0)
    rc = emu().emu_scale_rows(ctypes.byref(job), src.ctypes.data, stride_in, dst.ctypes.data, stride_out,
                              first, n, force_general, ctypes.byref(k))
    assert rc == 0
    return dst[:n * stride_out], k.value


def rows_view(buf, h, row_bytes, stride):
    return np.lib.stride_tricks.as_strided(buf, (h, row_bytes), (stride, 1))


def aligned_copy(buf, align=16):
    """Copy of `buf` whose first byte sits on an `align`-byte boundary."""
    raw = np.zeros(buf.size + align, np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + buf.size]
    out[:] = buf
    return out


def have_gpu():
    try:
        return sm.lib().smol_hip_device_count() > 0
    except Exception:
        return False
