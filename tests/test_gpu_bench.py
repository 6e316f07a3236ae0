"""-m gpu: `bench.py` end to end with a handful of steps -- the JSON contract the driver reads (one line, the last on stdout) and the fields
the measurement section of DESIGN.md defines, for the detector path (two lanes + the one-stream rate beside it) and for `--train`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1])                                   # the JSON line is the LAST line


def test_detector_line_has_the_contract_fields():
    j = _run("--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-fp32", "--no-seam2", "--no-sustained")
    assert j["metric"].startswith("768x768 images/s") and j["unit"] == "images/s" and j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "bf16" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["lanes"] == 2 and j["single_stream"]["value"] > 100 and j["value"] > 0.9 * j["single_stream"]["value"]
    assert abs(j["value"] - 8 * 6 / (j["ms_per_step"] * 6e-3)) < 0.01 * j["value"]
    rf = j["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    # the kernel instantiation with the largest share of the forward: the last FPN level (one 2 ms launch) or, if that ever gets faster
    # than their sum, one of the two fused MBConv-head instantiations (39 launches each)
    assert rf["kernel"].startswith(("conv3x3_wl1+top", "mbconv_slice")) and rf["bound"] == "mfma" and 0.1 < rf["frac"] < 0.6
    assert rf["kernel"].startswith("mbconv_slice") or rf["frac"] > 0.25


def test_one_lane_and_forced_process_group():
    j = _run("--steps", "4", "--warmup", "2", "--lanes", "1", "--no-cpu-baseline", "--no-fp32", "--no-seam2", "--no-sustained", "--no-profile",
             env={"FTC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert j["lanes"] == 1 and "single_stream" not in j and j["value"] > 100 and "RCCL all-gather" in j["config"]["workload"]


def test_train_line_with_a_forced_process_group():
    """BASELINE configs[4]: the `--train` line's contract fields, run through the N > 1 branches (DDP bucket segments, all-reduce on the
    communication stream, side-stream joins) on the one GPU there is: a one-rank RCCL group.  (Round 6: one subprocess instead of two --
    the plain `--train` line runs inside the default line's `train_step` record, which test_detector_line checks.)"""
    j = _run("--train", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", env={"FTC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29534"})
    assert j["metric"].startswith("768x768 images/s (train step") and j["finite"] is True and j["value"] > 20 and j["n_gpus"] == 1 and j["dtype"] == "bf16"
    assert j["plan_ops"] > 2000 and j["roofline"]["bound"] in ("mfma", "hbm") and 0 < j["roofline"]["frac"] < 1
