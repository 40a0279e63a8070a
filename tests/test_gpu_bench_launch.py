"""bench.py starts its own ranks: `python bench.py --gpus 2 ...` with no RANK in the environment must
launch two processes (before touching a GPU), run the batch-sharded loop of the reference's README
shape (src/main.rs:26-54) on each, and print ONE JSON line.  More ranks than visible GPUs is refused
(a record must not claim GPUs it did not run on) unless --allow-shared-devices is given: then, on a
one-GPU box, both ranks share device 0, line up over gloo, and the record says n_gpus = 1, ranks = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _visible_gpus():
    import torch

    return torch.cuda.device_count()


def _run(extra, expect_ok=True):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "mont16", "--steps", "2",
           "--warmup", "1", "--kernel-reps", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    if not expect_ok:
        return out
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _share():
    """Extra arguments and the n_gpus to expect for a two-rank run on this box."""
    return (["--allow-shared-devices"], 1) if _visible_gpus() < 2 else ([], 2)


def test_more_ranks_than_gpus_is_refused():
    if _visible_gpus() >= 2:
        pytest.skip("needs a box with one visible GPU")
    out = _run(["--batch", "16"], expect_ok=False)
    assert out.returncode != 0
    assert out.stdout.strip() == "" and "share devices" in out.stderr


def test_two_ranks_weak():
    extra, gpus = _share()
    rec = _run(["--batch", "64"] + extra)
    assert rec["n_gpus"] == gpus and rec["ranks"] == 2 and rec["scaling"] == "weak"
    assert ("share devices" in rec["config"]["sharding"]) == (gpus == 1)
    assert rec["config"]["batch_per_gpu"] == 64 and rec["config"]["job_batch"] == 128
    assert rec["value"] > 0 and rec["steps"] == 2
    assert abs(rec["value"] - 128 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 1e-3
    power = rec["roofline"]["power"]  # read from the card's hwmon files where the box exposes them
    assert power is None or (power["board_w"] > 0 and (power["limit_w"] is None or power["limit_w"] >= 100))


def test_two_ranks_strong():
    extra, gpus = _share()
    rec = _run(["--total-batch", "97"] + extra)  # ragged: 49 + 48
    assert rec["n_gpus"] == gpus and rec["ranks"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["job_batch"] == 97 and rec["config"]["batch_per_gpu"] == 49
