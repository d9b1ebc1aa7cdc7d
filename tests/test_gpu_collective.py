"""The N > 1 code path on the box it will run on, entered the way a bare `python bench.py --gpus N` enters it: no launcher
environment, bench.py starts its own rank(s) (self_launch -> torch.distributed.run), `init_process_group("nccl")` (RCCL),
bench.py's config-4 branch on the 1024-pair shard with the all_gather FORCED (not the world == 1 early return), the
gathered [B,26] rows (transform + 40-byte pair row) compared with the ungathered ones bit for bit.  The
8-GPU run of the driver is then not the first execution of that code."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_config4_branch_through_rccl_on_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--workload", "config4", "--pairs", "1024",
           "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-pairs", "0", "--force-collective", "--check-gather"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    chk = out["gather_check"]
    print(json.dumps({"gather_check": chk, "value": out["value"], "ms_per_step": out["ms_per_step"]}))
    assert chk["backend"] == "nccl"
    assert chk["rows"] == [1024, 26]
    assert chk["identical_to_local_rows"] and chk["pair_index_column_ok"]
    assert chk["rccl_library"] and "rccl" in chk["rccl_library"]
    assert out["n_gpus"] == 1 and out["config"]["pairs_total"] == 1024 and out["value"] > 0
