"""Developer tool: ten hist_icp calls of bench.py's ragged real-shape batch (argv[1] = matched | independent), for
rocprofv3 (tools/profile_ragged.sh).  Prints the ICP iteration count and the valid lengths' sums (the algorithmic bytes)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
B, N = 128, 10000
sizes = "matched" if (sys.argv[1:] or ["matched"])[0] == "matched" else True
S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=sizes, n_min=20)
ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=100, icp_stop_mode="reference")
REPS = int(os.environ.get("REPS", "10"))
T, it = utils_match.hist_icp(a, src, dst, return_iterations=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(REPS):
    T, it = utils_match.hist_icp(a, src, dst, return_iterations=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / REPS * 1e3
print(json.dumps({"sizes": "matched" if sizes == "matched" else "independent", "pairs": B, "padded": N, "calls": REPS + 1,
                  "ms_per_batch": round(ms, 3), "registrations_per_s": round(B / ms * 1e3, 1), "icp_iterations": int(it.item()),
                  "sum_valid_points": int(ns.sum() + nd.sum()), "sum_ns_times_nd": int((ns.astype(np.int64) * nd).sum()),
                  "library_build": _lib.BUILD_INFO}))
