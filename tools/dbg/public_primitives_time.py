import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_hist, hist as H, utils_helper
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0)
ex, ey, ez = utils_hist.bin_edges(a)
def f(): return H.hist(d, s, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
def g(): return utils_helper.nearest_neighbor_batch(s, d)
for name, fn in (("hist (public, all-pairs vote)", f), ("nearest_neighbor_batch", g)):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, "ms per 256x1024x1024 batch:", (time.perf_counter() - t) / 10 * 1e3)
