"""Developer tool: K different config-2 batches through hist_icp_many against a Python loop over S torch streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
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
ms = timeit(lambda: utils_match.hist_icp_many(a, S, D))
print(f"hist_icp_many, {K} batches in one call: {K * B / ms:.1f} k/s")
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    def loop():
        for k in range(K):
            with torch.cuda.stream(streams[k % ns]):
                utils_match.hist_icp(a, S[k], D[k])
    ms = timeit(loop)
    print(f"python loop over {ns} torch stream(s): {K * B / ms:.1f} k/s")
# the same loop with the fork / join structure of icpflow_hist_icp_many (events), from Python
ns = 4
streams = [torch.cuda.Stream(dev) for _ in range(ns)]
def forked():
    cur = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event(); fork.record(cur)
    for st in streams: st.wait_event(fork)
    for k in range(K):
        with torch.cuda.stream(streams[k % ns]):
            utils_match.hist_icp(a, S[k], D[k])
    for st in streams:
        ev = torch.cuda.Event(); ev.record(st); cur.wait_event(ev)
ms = timeit(forked)
print(f"python loop over 4 torch streams, forked from / joined into the current stream: {K * B / ms:.1f} k/s")
s0 = torch.cuda.Stream(dev)
with torch.cuda.stream(s0):
    ms = timeit(lambda: utils_match.hist_icp_many(a, S, D))
print(f"hist_icp_many called on a non-default stream: {K * B / ms:.1f} k/s")
