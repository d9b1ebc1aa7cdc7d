"""Developer tool: can one hist_icp call be captured into a HIP graph and replayed?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
B, N = 256, 1024
S, D, _ = synthetic.make_batch(B, N, seed=0)
src, dst = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
ref = utils_match.hist_icp(a, src, dst).clone()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): utils_match.hist_icp(a, src, dst)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = utils_match.hist_icp(a, src, dst)
torch.cuda.synchronize()
out.zero_()
g.replay(); torch.cuda.synchronize()
print("replay equals eager:", bool(torch.equal(out, ref)))
t = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print("graph replay ms/step", (time.perf_counter() - t) / 50 * 1e3)
t = time.perf_counter()
for _ in range(50): utils_match.hist_icp(a, src, dst)
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t) / 50 * 1e3)
