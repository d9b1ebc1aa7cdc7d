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
    # every N > 1 line names its own single-GPU point (the same workload on one GPU) and the efficiency against it
    assert "n1_same_workload" in out and out["n1_same_workload"]["pairs"] == 101 and "scaling_efficiency" in out


def test_self_launch_refuses_when_devices_are_missing():
    """More ranks than devices: a plain message about devices, not a launcher error (and nothing is spawned)."""
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert f"needs {n} devices, found {torch.cuda.device_count()}" in r.stderr, r.stderr[-2000:]
    assert "Traceback" not in r.stderr


def test_stream_workload_eight_gloo_ranks_uneven_shares():
    """`bench.py --gpus 8 --workload stream` -- 13 frame pairs dealt round-robin to eight self-launched
    ranks (shares 2,2,2,2,2,1,1,1), the one all_reduce per step, run_stream's summary; gloo, a stand-in for the registration
    (`--launch-selftest`).  One JSON line naming the stream, n_gpus and the per-rank counts."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--workload", "stream", "--frame-pairs", "13", "--steps", "2", "--warmup", "1",
                        "--launch-selftest"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["unit"] == "ms/frame-pair" and out["higher_is_better"] is False
    assert "frame-pair stream" in out["config"]["workload"]
    assert out["config"]["frame_pairs_per_gpu"] == [2, 2, 2, 2, 2, 1, 1, 1] and out["config"]["frame_pairs_total"] == 13
    assert out["reduce_check"] == {"frame_pairs_counted_by_all_ranks": 13, "equals_sum_of_shares": True, "backend": "gloo", "rccl_library": None}
    assert out["accuracy"]["frame_pairs"] == 13 and out["accuracy"]["epe"] == 0.0
    assert out["n1_same_workload"]["frame_pairs"] == 13 and out["n1_same_workload"]["unit"] == "ms/frame-pair" and out["scaling_efficiency"] is not None


def test_stream_workload_refuses_when_devices_are_missing():
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--workload", "stream", "--steps", "1"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    assert f"needs {n} devices, found {torch.cuda.device_count()}" in r.stderr, r.stderr[-2000:]
