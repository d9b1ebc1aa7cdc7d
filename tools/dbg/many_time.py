"""Developer tool: hist_icp_many on K config-2 batches against K separate calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
dev = torch.device("cuda", 0)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50)
B = 256
allb = [synthetic.make_batch(B, 1024, seed=0, first=k * B) for k in range(16)]
S = [torch.from_numpy(m[0]).to(dev) for m in allb]; D = [torch.from_numpy(m[1]).to(dev) for m in allb]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for K in (1, 2, 4, 8, 16):
    ms_many = timeit(lambda: utils_match.hist_icp_many(a, S[:K], D[:K]))
    ms_sep = timeit(lambda: [utils_match.hist_icp(a, s, d) for s, d in zip(S[:K], D[:K])])
    print(f"K {K}: one call {ms_many:.3f} ms = {K * B / ms_many:.1f} k/s; separate calls {ms_sep:.3f} ms = {K * B / ms_sep:.1f} k/s")
