import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match, _lib
dev = torch.device("cuda", 0)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50)
B, K = 256, 16
allb = [synthetic.make_batch(B, 1024, seed=0, first=k * B) for k in range(K)]
S = [torch.from_numpy(m[0]).to(dev) for m in allb]; D = [torch.from_numpy(m[1]).to(dev) for m in allb]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for kk in (4, 16):
    ms = timeit(lambda: utils_match.hist_icp_many(a, S[:kk], D[:kk]))
    print(f"K {kk}: hist_icp_many {kk * B / ms:.1f} k/s")
    with _lib.options(no_side_stream=True):
        ms = timeit(lambda: utils_match.hist_icp_many(a, S[:kk], D[:kk]))
    print(f"K {kk}: hist_icp_many without side streams {kk * B / ms:.1f} k/s")
