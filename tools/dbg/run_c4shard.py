"""Developer tool: a few registrations of BASELINE config 4's per-GPU shard (for profiler passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from icp_flow_amd import synthetic, utils_match
from oracle import reference_path as rp
B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 2048))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
for _ in range(3): utils_match.hist_icp(a, s, d)
torch.cuda.synchronize()
