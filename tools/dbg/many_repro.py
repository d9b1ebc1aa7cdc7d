"""Developer tool: four config-2 batches in ONE hist_icp_many call (bench.py extras: four_batches_in_one_call), timed in THIS process --
run it several times per setting of GPU_MAX_HW_QUEUES to see the modes (369 k vs 473 k registrations/s ten minutes apart)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
dev = torch.device("cuda", 0)
B, N = 256, 1024
args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
many = [synthetic.make_batch(B, N, seed=0, first=k * B) for k in range(4)]
srcs = [torch.from_numpy(m[0]).to(dev) for m in many]; dsts = [torch.from_numpy(m[1]).to(dev) for m in many]
utils_match.hist_icp_many(args, srcs, dsts); torch.cuda.synchronize()
res = []
for _ in range(5):
    t = time.perf_counter()
    for _ in range(10): utils_match.hist_icp_many(args, srcs, dsts)
    torch.cuda.synchronize(); res.append(4 * B * 10 / (time.perf_counter() - t))
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}: four batches in one call, k registrations/s: " + " ".join(f"{r / 1e3:.0f}" for r in res), flush=True)
