"""Developer tool: step time of config 2 and of config 4's shard with the library selected by ICPFLOW_HIP_LIB (vote variants)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match, utils_hist
dev = torch.device("cuda", 0)
out = []
for B, N, reps in ((256, 1024, 50), (1024, 2048, 8)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    best, besti = 1e9, 1e9
    for rep in range(4):
        for _ in range(3): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / reps * 1e3)
        for _ in range(3): utils_hist.estimate_init_pose(a, s, d)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): utils_hist.estimate_init_pose(a, s, d)
        torch.cuda.synchronize(); besti = min(besti, (time.perf_counter() - t) / reps * 1e3)
    out.append(f"{B}x{N}: step {best:.4f} ms, estimate_init_pose {besti:.4f} ms")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), " | ".join(out))
