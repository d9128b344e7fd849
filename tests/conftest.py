import os
import subprocess
import sys

import pytest
# torch first: it carries its own copy of the HIP runtime, and a process must settle on ONE copy.
# Loaded first, torch's is the one the product library then binds to (the order bench.py uses);
# the other way round torch finds no GPU.
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# No process-level memory hardening here (rounds 1-3 switched THP off for the test process): the tests hand ordinary
# pageable numpy memory to the host entry points, which is exactly what the library's bounce rings are for.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Oracle + product libraries are built in-tree; (re)build if missing."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libsmol_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    if not os.path.exists(os.path.join(ROOT, "smolscale_amd", "libsmolscale_amd.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smolscale_amd", "csrc")])
    yield
