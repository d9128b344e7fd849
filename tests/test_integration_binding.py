"""The reference-side binding compiled for real: the reference tree,
patched by tests/integration/apply_hip_binding.py + smolscale-hip.c, must pass exactly the
verify.c suites the unpatched reference passes -- without a GPU through the fall-through to the
CPU implementations, with a GPU through libsmolscale_amd.so (reference smolscale.c:507-546,
725-814, 816-876; verify.c:156-526)."""
import os
import subprocess

import pytest

import helpers as H

HERE = os.path.join(H.ROOT, "tests", "integration")
BUILD = os.path.join(HERE, "_build")


def _read(name):
    with open(os.path.join(BUILD, name)) as fh:
        return fh.read()


def test_patched_reference_builds_and_falls_through_without_a_gpu():
    if not os.path.exists("/root/reference/smolscale.c"):
        pytest.skip("reference sources not present (GPU box): the prebuilt binaries are tested by the gpu test")
    out = subprocess.run(["bash", os.path.join(HERE, "build_ref_with_hip.sh")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all three verify binaries agree" in out.stdout
    # the first three suites pass, the fourth fails on the reference itself -- identically everywhere
    ref = _read("verify_ref.out")
    assert "Ordering: ok" in ref and "Unassociated alpha: ok" in ref and "Saturation: ok" in ref
    assert _read("verify_patched_nohip.out") == ref and _read("verify_hip.out") == ref
    if not H.have_gpu():
        assert "no HIP device, using the CPU implementations" in _read("verify_hip.err")


@pytest.mark.gpu
@pytest.mark.skipif(not H.have_gpu(), reason="needs a HIP device")
def test_patched_reference_runs_verify_suites_on_the_gpu():
    exe = os.path.join(BUILD, "verify_hip")
    if not os.path.exists(exe):
        pytest.skip("tests/integration/_build/verify_hip not prebuilt")
    env = dict(os.environ, SMOL_HIP_VERBOSE="1", VERIFY_LINE_BUDGET="200000")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert "HIP backend (gfx950) active" in run.stderr
    want = _read("verify_ref_generic.out")          # recorded where the reference was built
    assert run.returncode == int(_read("verify_ref_generic.rc"))
    assert run.stdout == want
