"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import helpers as H
import smolscale_amd as sm

INCLUDE = os.path.join(H.ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(smol_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_loads_and_exports_every_declared_symbol():
    lib = ctypes.CDLL(sm.LIB_PATH)
    declared = declared_functions("smolscale.h") | declared_functions("smol_hip.h") | declared_functions("smol_cpu.h")
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), "declared in include/ but not exported: " + name
    # and the Python binding table covers exactly the declared set
    assert declared == set(sm.EXPORTS)


def test_struct_layout_matches_header():
    # SmolHipJob: 4*u32 + 12*u8 + 2*u32 + 4*u64
    assert ctypes.sizeof(sm.SmolHipJob) == 72
    assert sm.SmolHipJob.span_mul_x.offset == 28 and sm.SmolHipJob.frac0_x.offset == 40
    assert sm.SmolHipJob.frac0_x.offset % 8 == 0
    assert sm.SmolHipImage.src.offset == ctypes.sizeof(sm.SmolHipJob)


def test_invalid_arguments_are_rejected_without_a_gpu():
    job = sm.SmolHipJob()
    lib = sm.lib()
    assert lib.smol_hip_plan_job(10, 4, 4, 0, 4, 4, 0, ctypes.byref(job)) < 0
    assert lib.smol_hip_plan_job(0, 0, 4, 0, 4, 4, 0, ctypes.byref(job)) < 0
    assert lib.smol_hip_plan_job(0, 65536, 4, 0, 4, 4, 0, ctypes.byref(job)) < 0
    assert lib.smol_hip_plan_job(0, 65535, 1, 9, 1, 65535, 1, ctypes.byref(job)) == 0
    assert b"invalid" in lib.smol_hip_strerror(-10001)
    # smol_hip_memory_zones checks its arguments before it touches a device
    z = (ctypes.c_uint8 * 4)()
    assert lib.smol_hip_memory_zones(0, None, 4 << 30, 1 << 30, z, 4, None, None) == -10001
    assert lib.smol_hip_memory_zones(0, ctypes.c_void_p(1 << 40), 4 << 30, 256 << 20, z, 16, None, None) == -10001     # window below 512 MB
    assert lib.smol_hip_memory_zones(0, ctypes.c_void_p(1 << 40), 8 << 30, 1 << 30, z, 4, None, None) == -10001         # 8 windows, 4 labels
    assert lib.smol_hip_memory_zones(0, ctypes.c_void_p(1 << 40), 1 << 29, 1 << 30, z, 4, None, None) == -10001         # region below one window
    assert lib.smol_hip_memory_zones(0, ctypes.c_void_p((1 << 40) + 4), 4 << 30, 1 << 30, z, 4, None, None) == -10001   # base not 16-byte aligned
    assert lib.smol_hip_memory_zones(0, ctypes.c_void_p(1 << 40), 4 << 30, (1 << 30) + 8, z, 4, None, None) == -10001   # window not a multiple of 16


def test_no_product_code_touches_the_oracle():
    """The product tree must not include, link or load anything under oracle/."""
    pkg = os.path.join(H.ROOT, "smolscale_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("smol_oracle", "oracle/", "libsmolref", "_ref/"):
                    if needle == "oracle/" and f == "smol_luts.h":
                        continue        # generated banner mentions who verified it
                    assert needle not in text, "%s mentions %s" % (os.path.join(dirpath, f), needle)
    out = os.popen("ldd %s" % sm.LIB_PATH).read()
    assert "oracle" not in out and "smolref" not in out
