"""Developer tool: hist_icp calls of BASELINE config 4's per-GPU shard (argv[1] pairs x 2048 points, default 1024; <= 50 ICP
iterations), for rocprofv3 (tools/profile_workload.sh).  Prints one JSON line: ms per step, iteration count, library build."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
B = int((sys.argv[1:] or ["1024"])[0]); N = 2048
S, D, _ = synthetic.make_batch(B, N, seed=0)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50, icp_stop_mode="reference")
REPS = int(os.environ.get("REPS", "7"))
T, it = utils_match.hist_icp(a, src, dst, return_iterations=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(REPS):
    T, it = utils_match.hist_icp(a, src, dst, return_iterations=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / REPS * 1e3
print(json.dumps({"workload": f"config 4 shard: {B} pairs x {N} points, <= 50 ICP iterations", "pairs": B, "points": N, "calls": REPS + 1,
                  "ms_per_step": round(ms, 3), "registrations_per_s": round(B / ms * 1e3, 1), "icp_iterations": int(it.item()),
                  "library_build": _lib.BUILD_INFO}))
