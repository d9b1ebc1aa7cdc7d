"""bench.py --gpus N starts its own ranks when no launcher did: the launcher, the
rendezvous on 127.0.0.1, the contiguous shards and the one all_gather of [B,26] rows, driven over gloo on CPU tensors
(`--launch-selftest`: no GPU work, made-up rows), and the refusal on a box with too few devices."""
import json
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_self_launch_two_gloo_ranks_print_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--pairs", "101", "--launch-selftest"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2
    assert out["config"]["pairs_per_gpu"] == [51, 50] and out["config"]["pairs_total"] == 101
    assert out["gather_check"] == {"rows": [101, 26], "identical_to_all_ranks_rows": True, "backend": "gloo", "rccl_library": None}


def test_self_launch_refuses_when_devices_are_missing():
    """More ranks than devices: a plain message about devices, not a launcher error (and nothing is spawned)."""
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert f"needs {n} devices, found {torch.cuda.device_count()}" in r.stderr, r.stderr[-2000:]
    assert "Traceback" not in r.stderr
