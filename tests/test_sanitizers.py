"""AddressSanitizer + UBSan over the host-side product code and the emulated kernel bodies (SURVEY section 5:
the reference defines TEST_DEBUG_CFLAGS, Makefile:28, and never uses it). CPU container only -- GPU sanitizers are
not available on the pool. Each driver also compares every case with the oracle (built with the same sanitizers)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "tests", "asan")
CASES_CPU = os.environ.get("SMOL_ASAN_CASES_CPU", "120")
CASES_EMU = os.environ.get("SMOL_ASAN_CASES_EMU", "40")

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or os.environ.get("SMOL_SKIP_ASAN") == "1",
                                reason="needs the host compilers (build container)")


def _run(target, make_dir, make_target, cases, seed):
    subprocess.check_call(["make", "-s", "-C", make_dir, make_target])
    exe = os.path.join(ASAN, "_build", target)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, cases, seed], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "0 mismatches, no sanitizer report" in p.stdout, p.stdout[-2000:]


def test_host_code_under_asan_ubsan():
    """smol_plan.c + smol_cpu.c + smolscale_api.c (SMOL_BACKEND=cpu), `make -C smolscale_amd/csrc cpu-asan'"""
    _run("asan_cpu", os.path.join(ROOT, "smolscale_amd", "csrc"), "cpu-asan", CASES_CPU, "3")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++")
def test_emulated_kernel_bodies_under_asan_ubsan():
    """the kernel bodies of smolscale_amd/csrc/*.hip.h compiled for the host, `make -C tests/emu asan'"""
    _run("asan_emu", os.path.join(ROOT, "tests", "emu"), "asan", CASES_EMU, "4")
