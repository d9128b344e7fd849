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
    assert line["verified_vs_oracle"] == {"all_ranks_ok": True, "ranks_that_checked": 1}
    assert set(line["roofline"]["step_us_each"]) == {"min", "median", "mean"}
    assert "traffic_source" in line["roofline"]
    # round 6: the arenas are placed by probing (every probe reported) and the timed launches rotate over destination arenas -- 512 MB of
    # output between two writes of a line -- with the one-arena figure of rounds 1-5 beside
    assert r["destination_arenas"] == 4 and r["same_destination"]["launch_us"] > 0 and 0.3 < r["same_destination"]["frac"] < 1.0
    pl = line["placement"]
    assert pl["dst_rotation"] == 4 and len(pl["src_windows_us"]) >= 2 and len(pl["dst_windows_us"]) >= 2 and pl["picked_us"] > 0
    # the library's own labels of the region's windows (smol_hip_memory_zones) beside the probes: where the region spans more than
    # one kind of memory, the probes pick a source and a destination of different kinds
    zn = pl["zones"]
    assert len(zn["labels"]) == pl["region_GB"] * 1024 // zn["window_MB"] and zn["kinds"] >= 1
    if zn["kinds"] > 1 and len(zn["src"]) == 1 and len(zn["dst"]) == 1:
        assert zn["src"] != zn["dst"], pl
    assert "destination arenas" in line["config"]["workload"]
    rows = line["other_baseline_configs"]
    assert len(rows) == 7 and all("error" not in r for r in rows.values()), rows
    assert any(k.startswith("BOX") for k in rows) and any(k.startswith("C5mix") for k in rows)
    assert any(k.startswith("C2CROP") for k in rows)        # the alignment cliff is a measured row now
    # the drop-in API on host memory (PCIe inclusive) is stated in the line, beside cpu_baseline
    e = line["e2e"]
    for mode in ("pinned", "pageable"):
        assert e[mode]["bit_exact_vs_oracle"] is True, e
        assert e[mode]["one_caller"]["Mpix_out_per_s"] > 0 and e[mode]["16_callers"]["Mpix_out_per_s"] > 0


def _torchrun(n, port, *extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(H.ROOT, "bench.py"),
           "--gpus", str(n), "--steps", "5", "--warmup", "2", "--share-devices"] + list(extra)
    out = subprocess.check_output(cmd, text=True, stderr=subprocess.DEVNULL, timeout=900, env=env)
    return _last_json_line(out)


def test_two_ranks_share_the_gpu_and_report_the_aggregate():
    """Default --backend auto: RCCL cannot come up with two ranks on one GPU, so the run must agree
    on gloo by itself; every rank checks its own strip against the oracle."""
    import bench
    line = _torchrun(2, 29517)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["frames_per_step"] == 2 * bench.FRAMES_PER_GPU and line["config"]["rows_per_rank"] == 512
    assert line["value"] > 0
    assert line["barrier_backend"] in ("gloo", "nccl")
    assert line["verified_vs_oracle"] == {"all_ranks_ok": True, "ranks_that_checked": 2}


def test_three_ranks_uneven_strips():
    line = _torchrun(3, 29519, "--backend", "gloo", "--frames-per-gpu", "4")
    assert line["n_gpus"] == 3 and line["config"]["rows_per_rank"] == 342
    assert line["verified_vs_oracle"] == {"all_ranks_ok": True, "ranks_that_checked": 3}


def test_config4_rows_sharded_with_end_to_end_numbers():
    line = _torchrun(2, 29521, "--backend", "gloo", "--config", "c4")
    assert line["scaling"] == "strong" and line["config"]["rows_per_rank"] == 2048
    assert line["verified_vs_oracle"] == {"all_ranks_ok": True, "ranks_that_checked": 2}
    e = line["e2e"]
    for mode in ("pinned", "pinned_hostalloc", "pageable"):
        assert e[mode]["equal_to_device_resident"] is True and e[mode]["ms"] > 0
    assert e["bytes_over_pcie"] == 2048 * (65535 * 3 + 8192 * 4)
    # PCIe-inclusive time can only be longer than the kernel-only step
    assert e["pinned"]["ms"] > line["ms_per_step"]


def test_config5_mixed_thumbnails_sharded_by_image():
    line = _torchrun(2, 29523, "--backend", "gloo", "--config", "c5mix")
    assert line["scaling"] == "strong" and line["config"]["images_per_rank"] == 5000
    assert line["verified_vs_oracle"] == {"all_ranks_ok": True, "ranks_that_checked": 2}
    assert line["images_per_s"] > 0
