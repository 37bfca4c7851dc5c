"""bench.py's N > 1 control flow on CPU: `python bench.py --gpus 2 --backend gloo --dry-run` re-executes
itself under torch.distributed.run on 127.0.0.1, runs settle / warm-up / the timed regions with barriers and the MAX over
ranks, the double-buffered asynchronous all_gather_into_tensor, and prints ONE JSON line on rank 0 - with a stand-in
extractor on CPU tensors, so that the path the driver launches on an 8-GPU node has been executed end to end before it gets
there.  What the gather delivered is checked inside the run (`gather_verified`)."""

import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _run(gpus, launcher):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--dry-run", "--steps", "6", "--warmup", "2",
           "--batch", "48", "--frames", "50", "--min-seconds", "0.05", "--settle-seconds", "0.02"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", "29731"] + cmd[1:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                         # exactly one JSON line, from rank 0
    assert res.stdout.rstrip("\n").splitlines()[-1] == lines[0]            # ... and it is the LAST stdout line (what the driver parses)
    assert len(lines[0]) <= 4096, len(lines[0])                # a 19 KB line was not held by the driver
    assert "bench.py full record: {" in res.stderr              # the long record goes to stderr and gpurun_out/bench_full.json
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus,launcher", [(1, False), (2, False), (2, True), (3, False)])
def test_bench_dry_run_control_flow(gpus, launcher):
    rec = _run(gpus, launcher)
    for k in REQUIRED:
        assert k in rec, k
    assert rec["n_gpus"] == gpus and rec["steps"] == 6 and rec["warmup"] == 2 and rec["dry_run"] is True and rec["scaling"] == "weak"
    assert rec["config"]["global_batch_utts"] == gpus * 48 and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert abs(rec["value"] - gpus * 48 / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 0.01        # value = whole-job units / time
    if gpus > 1:
        assert rec["gather_verified"] is True and rec["backend"] == "gloo"


def test_bench_refuses_gloo_without_dry_run():
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--backend", "gloo"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "--dry-run" in (res.stdout + res.stderr)


def test_compact_line_of_a_full_device_record_stays_small():
    """The compact line built from a FULL single-GPU record (every supplementary model / mode, the gate tables, the ark -> ark runs:
    profiles/r5y_bench.json is the 19.4 KB line the round-5 driver could not hold) keeps the contract's keys, the dominant kernel's
    roofline, the CPU baseline and one scalar per supplementary record inside 4 KB."""
    sys.path.insert(0, REPO)
    import bench
    with open(os.path.join(REPO, "profiles", "r5y_bench.json")) as f:
        full = json.loads(f.read())
    rec = bench.compact_record(full)
    line = json.dumps(rec, separators=(",", ":"))
    assert len(line) <= bench.LINE_TARGET, len(line)
    for k in REQUIRED + ("roofline", "cpu_baseline", "value_parity_grade", "value_single_stream", "supplementary"):
        assert k in rec, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_us", "dominant_frac", "dominant_flop_per_launch", "traffic_over_algorithmic"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    for k in ("ecapa_c3", "ecapa_c3_f32x", "resnet_c5", "resnet_c5_f32x", "xvector_f32x", "ark_stream_f32x", "ark_sharded_f32x"):        # (r5y: the record of the round before f32m)
        assert k in rec["supplementary"], k
    assert all(not isinstance(v, list) or len(v) <= 2 for sub in rec["supplementary"].values() for v in sub.values())
