"""smolscale_amd -- MI355X-native image resampler with smolscale's C API.

The product is the shared library ``libsmolscale_amd.so`` (host C planner + HIP
kernels for gfx950, sources in ``smolscale_amd/csrc``). This Python package is
only the host-side binding used by the tests and by ``bench.py``: thin ctypes
mirrors of the C entry points with the same names and argument meaning as
``include/smolscale.h`` (the reference's public header, smolscale.h:48-83) and
``include/smol_hip.h`` (the C-ABI boundary of the HIP backend).

There is no CPU implementation behind any of it: if the library is missing,
importing :func:`lib` raises; if the GPU is missing, the C API aborts the process
with a message (the reference aborts on its own unusable configurations,
smolscale.c:779-780).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOLSCALE_AMD_LIB") or os.path.join(_HERE, "libsmolscale_amd.so")

# SmolPixelType, include/smolscale.h
SMOL_PIXEL_RGBA8_PREMULTIPLIED = 0
SMOL_PIXEL_BGRA8_PREMULTIPLIED = 1
SMOL_PIXEL_ARGB8_PREMULTIPLIED = 2
SMOL_PIXEL_ABGR8_PREMULTIPLIED = 3
SMOL_PIXEL_RGBA8_UNASSOCIATED = 4
SMOL_PIXEL_BGRA8_UNASSOCIATED = 5
SMOL_PIXEL_ARGB8_UNASSOCIATED = 6
SMOL_PIXEL_ABGR8_UNASSOCIATED = 7
SMOL_PIXEL_RGB8 = 8
SMOL_PIXEL_BGR8 = 9
SMOL_PIXEL_MAX = 10

SMOL_HIP_FILTER_COPY, SMOL_HIP_FILTER_ONE, SMOL_HIP_FILTER_BILINEAR, SMOL_HIP_FILTER_BOX = 0, 1, 2, 3

_u8 = ctypes.c_uint8
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p
_int = ctypes.c_int


def bytes_per_pixel(pixel_type):
    return 3 if pixel_type >= SMOL_PIXEL_RGB8 else 4


class SmolHipJob(ctypes.Structure):
    """include/smol_hip.h: SmolHipJob"""
    _fields_ = [
        ("width_in", _u32), ("height_in", _u32), ("width_out", _u32), ("height_out", _u32),
        ("pixel_type_in", _u8), ("pixel_type_out", _u8),
        ("filter_h", _u8), ("filter_v", _u8),
        ("halvings_h", _u8), ("halvings_v", _u8),
        ("wide", _u8), ("premul16", _u8), ("linear", _u8), ("quirk24", _u8),
        ("reserved", _u8 * 2),
        ("span_mul_x", _u32), ("span_mul_y", _u32),
        ("frac0_x", _u64), ("step_x", _u64), ("frac0_y", _u64), ("step_y", _u64),
    ]


class SmolHipImage(ctypes.Structure):
    """include/smol_hip.h: SmolHipImage"""
    _fields_ = [
        ("job", SmolHipJob), ("src", _vp), ("dst", _vp),
        ("rowstride_in", _u32), ("rowstride_out", _u32),
        ("first_row", _u32), ("n_rows", _u32),
    ]


# SmolPostRowFunc (include/smolscale.h)
POST_ROW_FUNC = ctypes.CFUNCTYPE(None, ctypes.POINTER(_u32), _int, _vp)

_SCALE_ARGS = [_vp, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _u8]

# name -> (restype, argtypes): every symbol include/*.h declares
EXPORTS = {
    # include/smolscale.h
    "smol_scale_simple": (None, _SCALE_ARGS),
    "smol_scale_new": (_vp, _SCALE_ARGS),
    "smol_scale_new_full": (_vp, _SCALE_ARGS + [_vp, _vp]),
    "smol_scale_destroy": (None, [_vp]),
    "smol_scale_batch": (None, [_vp, _u32, _u32]),
    "smol_scale_batch_full": (None, [_vp, _vp, _u32, _u32]),
    # include/smol_hip.h
    "smol_hip_plan_job": (_int, [_u32] * 7 + [ctypes.POINTER(SmolHipJob)]),
    "smol_hip_input_row_range": (_int, [ctypes.POINTER(SmolHipJob), _u32, _u32,
                                        ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "smol_hip_bilinear_sample": (None, [_u64, _u64, _u32, _u32, ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "smol_hip_box_span": (None, [_u64, _u32, _u32] + [ctypes.POINTER(_u32)] * 4),
    "smol_hip_strerror": (ctypes.c_char_p, [_int]),
    "smol_hip_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(SmolHipJob), _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_hip_device_count": (_int, []),
    "smol_hip_plan_create": (_int, [ctypes.POINTER(SmolHipJob), _int, ctypes.POINTER(_vp)]),
    "smol_hip_plan_destroy": (None, [_vp]),
    "smol_hip_plan_prepare": (_int, [_vp]),
    "smol_hip_plan_reserve": (_int, [_vp, _u32]),
    "smol_hip_scale_rows_host": (_int, [_vp, _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_hip_scale_rows_host_multi": (_int, [ctypes.POINTER(_vp), _u32, _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_hip_scale_rows_peer": (_int, [_vp, _int, _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_hip_host_register": (_int, [_vp, _u64]),
    "smol_hip_host_unregister": (_int, [_vp]),
    "smol_hip_host_alloc": (_int, [_u64, ctypes.POINTER(_vp)]),
    "smol_hip_host_range_is_pinned": (_int, [_vp, _u64]),
    "smol_hip_plan_resident_source": (_int, [_vp, _int]),
    "smol_hip_host_free": (_int, [_vp]),
    "smol_hip_scale_rows_device": (_int, [_vp, _vp, _u32, _vp, _u32, _u32, _u32, _vp]),
    "smol_hip_batch_create": (_int, [ctypes.POINTER(SmolHipImage), _u32, _int, ctypes.POINTER(_vp)]),
    "smol_hip_batch_launch": (_int, [_vp, _vp]),
    "smol_hip_batch_use_graph": (_int, [_vp, _int]),
    "smol_hip_batch_info": (_int, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_char_p)]),
    "smol_hip_batch_destroy": (None, [_vp]),
    "smol_hip_batch_time": (_int, [_vp, _u32, _u32, ctypes.POINTER(ctypes.c_float)]),
    "smol_hip_batch_time_each": (_int, [_vp, _u32, _u32, ctypes.POINTER(ctypes.c_float)]),
    "smol_hip_batch_time_rotating": (_int, [ctypes.POINTER(_vp), _u32, _u32, _u32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "smol_hip_memory_zones": (_int, [_int, _vp, _u64, _u64, ctypes.POINTER(ctypes.c_uint8), _u32, ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_float)]),
    "smol_hip_batch_create_multi": (_int, [ctypes.POINTER(SmolHipImage), _u32, ctypes.POINTER(_int), _u32,
                                           ctypes.POINTER(_vp)]),
    "smol_hip_batch_multi_launch": (_int, [_vp]),
    "smol_hip_batch_multi_sync": (_int, [_vp]),
    "smol_hip_batch_multi_time": (_int, [_vp, _u32, _u32, ctypes.POINTER(ctypes.c_float)]),
    "smol_hip_batch_multi_destroy": (None, [_vp]),
    "smol_hip_shutdown": (None, []),
    # include/smol_cpu.h
    "smol_cpu_scale_rows": (_int, [ctypes.POINTER(SmolHipJob), _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_cpu_plan_create": (_int, [ctypes.POINTER(SmolHipJob), ctypes.POINTER(_vp)]),
    "smol_cpu_plan_scale_rows": (_int, [_vp, _vp, _u32, _vp, _u32, _u32, _u32]),
    "smol_cpu_plan_destroy": (None, [_vp]),
    "smol_hip_malloc": (_int, [_int, _u64, ctypes.POINTER(_vp)]),
    "smol_hip_free": (_int, [_int, _vp]),
    "smol_hip_memcpy_h2d": (_int, [_int, _vp, _vp, _u64]),
    "smol_hip_memcpy_d2h": (_int, [_int, _vp, _vp, _u64]),
    "smol_hip_memset": (_int, [_int, _vp, _int, _u64]),
    "smol_hip_device_synchronize": (_int, [_int]),
}

_lib = None


def build():
    """Compile libsmolscale_amd.so in-tree (hipcc --offload-arch=gfx950)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    """The loaded C library. Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "smolscale_amd: %s is missing; build it with `make -C smolscale_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in EXPORTS.items():
            fn = getattr(handle, name)          # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class HipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise HipError("%s failed: %s (%d)" % (what or "HIP call", lib().smol_hip_strerror(rc).decode(), rc))


def plan_job(pixel_type_in, width_in, height_in, pixel_type_out, width_out, height_out, with_srgb=0):
    job = SmolHipJob()
    check(lib().smol_hip_plan_job(pixel_type_in, width_in, height_in, pixel_type_out,
                                  width_out, height_out, int(with_srgb), ctypes.byref(job)),
          "smol_hip_plan_job")
    return job


def kernel_name(job, src_ptr=256, rowstride_in=None, dst_ptr=256, rowstride_out=None, first_row=0, n_rows=None):
    """smol_hip_kernel_name(): the kernel family the dispatcher picks for these rows (default: aligned, packed rows,
    the whole image)."""
    if rowstride_in is None:
        rowstride_in = job.width_in * bytes_per_pixel(job.pixel_type_in)
    if rowstride_out is None:
        rowstride_out = job.width_out * bytes_per_pixel(job.pixel_type_out)
    if n_rows is None:
        n_rows = job.height_out - first_row
    return lib().smol_hip_kernel_name(ctypes.byref(job), src_ptr, rowstride_in, dst_ptr, rowstride_out,
                                      first_row, n_rows).decode()


def input_row_range(job, first_row, n_rows):
    a, b = _u32(0), _u32(0)
    check(lib().smol_hip_input_row_range(ctypes.byref(job), first_row, n_rows,
                                         ctypes.byref(a), ctypes.byref(b)), "smol_hip_input_row_range")
    return a.value, b.value


def _as_u8(buf):
    arr = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return arr


def smol_scale_simple(pixels_in, pixel_type_in, width_in, height_in, rowstride_in,
                      pixels_out, pixel_type_out, width_out, height_out, rowstride_out, with_srgb=0):
    """smol_scale_simple() of include/smolscale.h on numpy uint8 buffers (in place)."""
    src = _as_u8(pixels_in)
    dst = pixels_out.view(np.uint8).reshape(-1)
    lib().smol_scale_simple(src.ctypes.data, pixel_type_in, width_in, height_in, rowstride_in,
                            dst.ctypes.data, pixel_type_out, width_out, height_out, rowstride_out,
                            int(with_srgb))


class SmolScaleCtx:
    """smol_scale_new() / smol_scale_batch() / smol_scale_destroy() of include/smolscale.h."""

    def __init__(self, pixels_in, pixel_type_in, width_in, height_in, rowstride_in,
                 pixels_out, pixel_type_out, width_out, height_out, rowstride_out,
                 with_srgb=0, post_row_func=None, user_data=None):
        self._src = _as_u8(pixels_in)
        self._dst = None if pixels_out is None else pixels_out.view(np.uint8).reshape(-1)
        self._cb = POST_ROW_FUNC(post_row_func) if post_row_func else None
        dst_ptr = None if self._dst is None else self._dst.ctypes.data
        if self._cb:
            self._ctx = lib().smol_scale_new_full(
                self._src.ctypes.data, pixel_type_in, width_in, height_in, rowstride_in,
                dst_ptr, pixel_type_out, width_out, height_out, rowstride_out, int(with_srgb),
                ctypes.cast(self._cb, _vp), user_data)
        else:
            self._ctx = lib().smol_scale_new(
                self._src.ctypes.data, pixel_type_in, width_in, height_in, rowstride_in,
                dst_ptr, pixel_type_out, width_out, height_out, rowstride_out, int(with_srgb))
        self.rowstride_out = rowstride_out

    def batch(self, first_outrow, n_outrows):
        lib().smol_scale_batch(self._ctx, first_outrow, n_outrows)

    def batch_full(self, outrows_dest, first_outrow, n_outrows):
        dst = outrows_dest.view(np.uint8).reshape(-1)
        lib().smol_scale_batch_full(self._ctx, dst.ctypes.data, first_outrow, n_outrows)

    def destroy(self):
        if self._ctx:
            lib().smol_scale_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()


class HipBatch:
    """smol_hip_batch_*() of include/smol_hip.h: many images (device pointers), few launches."""

    def __init__(self, images, device=0):
        arr = (SmolHipImage * len(images))(*images)
        self._h = _vp()
        check(lib().smol_hip_batch_create(arr, len(images), device, ctypes.byref(self._h)),
              "smol_hip_batch_create")

    def launch(self, stream=None):
        check(lib().smol_hip_batch_launch(self._h, stream), "smol_hip_batch_launch")

    def use_graph(self, enable=True):
        check(lib().smol_hip_batch_use_graph(self._h, int(enable)), "smol_hip_batch_use_graph")

    def time(self, warmup, iters):
        ms = ctypes.c_float(0)
        check(lib().smol_hip_batch_time(self._h, warmup, iters, ctypes.byref(ms)), "smol_hip_batch_time")
        return ms.value

    def time_each(self, warmup, iters):
        """Per-launch durations in ms (one hipEvent between consecutive launches)."""
        ms = (ctypes.c_float * iters)()
        check(lib().smol_hip_batch_time_each(self._h, warmup, iters, ms), "smol_hip_batch_time_each")
        return list(ms)

    def info(self):
        n = _u32(0)
        name = ctypes.c_char_p()
        check(lib().smol_hip_batch_info(self._h, ctypes.byref(n), ctypes.byref(name)), "smol_hip_batch_info")
        return n.value, (name.value or b"").decode()

    def destroy(self):
        if self._h:
            lib().smol_hip_batch_destroy(self._h)
            self._h = None


class HipBatchRotation:
    """smol_hip_batch_time_rotating(): several batches of one device -- the same work into several destination arenas -- launched in
    rotation on one stream. Quacks like HipBatch for timing (time / time_each / info / destroy); batches [0] is the one whose outputs
    a checker reads."""

    def __init__(self, batches):
        self.batches = list(batches)
        self._arr = (_vp * len(self.batches))(*[b._h for b in self.batches])

    def _run(self, warmup, iters, each):
        ms = ctypes.c_float(0)
        arr = (ctypes.c_float * iters)() if each else None
        check(lib().smol_hip_batch_time_rotating(self._arr, len(self.batches), warmup, iters, ctypes.byref(ms), arr),
              "smol_hip_batch_time_rotating")
        return ms.value, (list(arr) if each else None)

    def time(self, warmup, iters):
        return self._run(warmup, iters, False)[0]

    def time_each(self, warmup, iters):
        return self._run(warmup, iters, True)[1]

    def info(self):
        return self.batches[0].info()

    def launch(self, stream=None):
        for b in self.batches:
            b.launch(stream)

    def use_graph(self, enable=True):
        for b in self.batches:
            b.use_graph(enable)

    def destroy(self):
        for b in self.batches:
            b.destroy()
        self.batches = []


def memory_zones(base, n_bytes, window, device=0):
    """smol_hip_memory_zones(): (labels, kinds, contrast) for the windows of `window' bytes (>= 512 MB) of the device region
    [base, base + n_bytes) -- a source arena and a destination arena in windows with DIFFERENT labels make a fast pair for a
    streaming kernel; sources that one launch reads together belong in windows with the SAME label (include/smol_hip.h)."""
    n = int(n_bytes) // int(window)
    z = (ctypes.c_uint8 * max(n, 1))()
    k = _u32(0)
    c = ctypes.c_float(0)
    check(lib().smol_hip_memory_zones(device, _vp(int(base)), int(n_bytes), int(window), z, n, ctypes.byref(k), ctypes.byref(c)),
          "smol_hip_memory_zones")
    return list(z)[:n], k.value, c.value


class HipBatchMulti:
    """smol_hip_batch_*_multi(): image i runs on devices[i % len(devices)] (device pointers there)."""

    def __init__(self, images, devices):
        arr = (SmolHipImage * len(images))(*images)
        devs = (_int * len(devices))(*devices)
        self.n_devices = len(devices)
        self._h = _vp()
        check(lib().smol_hip_batch_create_multi(arr, len(images), devs, len(devices), ctypes.byref(self._h)),
              "smol_hip_batch_create_multi")

    def launch(self):
        check(lib().smol_hip_batch_multi_launch(self._h), "smol_hip_batch_multi_launch")

    def sync(self):
        check(lib().smol_hip_batch_multi_sync(self._h), "smol_hip_batch_multi_sync")

    def time(self, warmup, iters):
        ms = (ctypes.c_float * self.n_devices)()
        check(lib().smol_hip_batch_multi_time(self._h, warmup, iters, ms), "smol_hip_batch_multi_time")
        return list(ms)

    def destroy(self):
        if self._h:
            lib().smol_hip_batch_multi_destroy(self._h)
            self._h = None


class HipPlan:
    """smol_hip_plan_*() and the row-range entries of include/smol_hip.h."""

    def __init__(self, job, device=0):
        self.job = job
        self.device = device
        self._h = _vp()
        check(lib().smol_hip_plan_create(ctypes.byref(job), device, ctypes.byref(self._h)), "smol_hip_plan_create")

    def prepare(self):
        check(lib().smol_hip_plan_prepare(self._h), "smol_hip_plan_prepare")

    def resident_source(self, on=True):
        check(lib().smol_hip_plan_resident_source(self._h, int(on)), "smol_hip_plan_resident_source")

    def reserve(self, images_in_flight):
        check(lib().smol_hip_plan_reserve(self._h, images_in_flight), "smol_hip_plan_reserve")

    def scale_rows_host(self, src, rowstride_in, dst, rowstride_out, first_row, n_rows):
        check(lib().smol_hip_scale_rows_host(self._h, _as_u8(src).ctypes.data, rowstride_in,
                                             dst.ctypes.data, rowstride_out, first_row, n_rows),
              "smol_hip_scale_rows_host")

    def scale_rows_device(self, src_ptr, rowstride_in, dst_ptr, rowstride_out, first_row, n_rows, stream=None):
        check(lib().smol_hip_scale_rows_device(self._h, src_ptr, rowstride_in, dst_ptr, rowstride_out,
                                               first_row, n_rows, stream), "smol_hip_scale_rows_device")

    def scale_rows_peer(self, src_device, src_ptr, rowstride_in, dst, rowstride_out, first_row, n_rows):
        check(lib().smol_hip_scale_rows_peer(self._h, src_device, src_ptr, rowstride_in,
                                             dst.ctypes.data, rowstride_out, first_row, n_rows),
              "smol_hip_scale_rows_peer")

    def destroy(self):
        if self._h:
            lib().smol_hip_plan_destroy(self._h)
            self._h = None


def scale_rows_host_multi(plans, src, rowstride_in, dst, rowstride_out, first_row, n_rows):
    """smol_hip_scale_rows_host_multi(): one output-row strip per plan (device), all in flight together."""
    arr = (_vp * len(plans))(*[p._h for p in plans])
    check(lib().smol_hip_scale_rows_host_multi(arr, len(plans), _as_u8(src).ctypes.data, rowstride_in,
                                               dst.ctypes.data, rowstride_out, first_row, n_rows),
          "smol_hip_scale_rows_host_multi")


class PinnedArray:
    """A page-locked numpy uint8 buffer: smol_hip_host_alloc() (hipHostMalloc), or -- register=True -- an anonymous
    mapping of the caller's registered with smol_hip_host_register()."""

    def __init__(self, n_bytes, register=False):
        n = int(n_bytes)
        self._registered = False
        self._ptr = None
        if register:
            import mmap
            self._map = mmap.mmap(-1, n + 4096)
            self.array = np.frombuffer(self._map, np.uint8, n)
            check(lib().smol_hip_host_register(self.array.ctypes.data, n), "smol_hip_host_register")
            self._registered = True
        else:
            p = _vp()
            check(lib().smol_hip_host_alloc(n, ctypes.byref(p)), "smol_hip_host_alloc")
            self._ptr = p.value
            self.array = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p.value))
            self.array[:] = 0

    def free(self):
        if self._registered:
            lib().smol_hip_host_unregister(self.array.ctypes.data)
            self._registered = False
            self.array = None
        if self._ptr:
            self.array = None
            lib().smol_hip_host_free(self._ptr)
            self._ptr = None


class DeviceBuffer:
    """Raw device memory through smol_hip_malloc()/free() (for hosts without a HIP binding)."""

    def __init__(self, n_bytes, device=0):
        self.device = device
        self.n_bytes = int(n_bytes)
        p = _vp()
        check(lib().smol_hip_malloc(device, self.n_bytes, ctypes.byref(p)), "smol_hip_malloc")
        self.ptr = p.value

    def upload(self, host, offset=0):
        h = _as_u8(host)
        check(lib().smol_hip_memcpy_h2d(self.device, self.ptr + offset, h.ctypes.data, h.size), "h2d")

    def download(self, n_bytes=None, offset=0):
        n = self.n_bytes - offset if n_bytes is None else n_bytes
        out = np.empty(n, np.uint8)
        check(lib().smol_hip_memcpy_d2h(self.device, out.ctypes.data, self.ptr + offset, n), "d2h")
        return out

    def fill(self, value):
        check(lib().smol_hip_memset(self.device, self.ptr, value, self.n_bytes), "memset")

    def free(self):
        if self.ptr:
            lib().smol_hip_free(self.device, self.ptr)
            self.ptr = None


def upload(dev_ptr, host, device=0):
    """host (numpy) -> device pointer through smol_hip_memcpy_h2d(): pageable memory travels through the library's
    page-locked bounce ring, never through the runtime's in-place pinning (also for tensors torch allocated)."""
    h = _as_u8(host)
    check(lib().smol_hip_memcpy_h2d(device, dev_ptr, h.ctypes.data, h.size), "h2d")


def download(dev_ptr, n_bytes, device=0):
    """device pointer -> a new numpy uint8 array through smol_hip_memcpy_d2h()."""
    out = np.empty(int(n_bytes), np.uint8)
    check(lib().smol_hip_memcpy_d2h(device, out.ctypes.data, dev_ptr, int(n_bytes)), "d2h")
    return out


# --------------------------------------------------------------------------- #
# Row-range sharding (multi-GPU): no data-path collective, each rank owns a    #
# strip of output rows and the slab of input rows that strip depends on.       #
# --------------------------------------------------------------------------- #

def strip_rows(rank, world, height_out):
    """Output rows of `rank`: the reference's MT strip rule (test.c:870-877),
    ceil(height_out / world) rows per worker, last strips may be short or empty."""
    rows = (height_out + world - 1) // world
    first = min(rank * rows, height_out)
    return first, max(0, min(rows, height_out - first))


def shard(job, rank, world):
    """(first_row, n_rows, in_row_first, in_row_last) for `rank`; n_rows may be 0."""
    first, n = strip_rows(rank, world, job.height_out)
    if n == 0:
        return first, 0, 0, -1
    r0, r1 = input_row_range(job, first, n)
    return first, n, r0, r1
