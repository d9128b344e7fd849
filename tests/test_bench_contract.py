"""bench.py keeps the driver's contract: one JSON line with the agreed keys, the
roofline and cpu_baseline objects, and a multi-rank launch (two ranks sharing the
one GPU of the test box, gloo for the barrier) shards rows correctly."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json_line(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line():
    out = subprocess.check_output([sys.executable, os.path.join(H.ROOT, "bench.py"), "--steps", "5", "--warmup", "2"],
                                  text=True, stderr=subprocess.DEVNULL, timeout=600)
    line = _last_json_line(out)
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["dtype"] == "u8" and line["vs_baseline"] is None
    assert "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.3 < r["frac"] < 1.0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert line["bit_exact_vs_reference"] is True


def test_two_ranks_share_the_gpu_and_report_the_aggregate():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(H.ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "2", "--backend", "gloo", "--share-devices"]
    out = subprocess.check_output(cmd, text=True, stderr=subprocess.DEVNULL, timeout=900, env=env)
    line = _last_json_line(out)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["frames_per_step"] == 16 and line["config"]["rows_per_rank"] == 512
    assert line["value"] > 0
