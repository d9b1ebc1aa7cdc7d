"""Developer tool: is the stream of demo frame pairs (4 in flight) bound by the host thread?  The same stream with the host's
sanity_grid answered from a cache (its 0.14 ms of numpy removed): if the figure moves by about that much, it is."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs, utils_match
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
a = frame_pairs.default_args(max_points=10000)
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
copies = [fp] * 12
def stream(k):
    for _ in frame_pairs.register_in_flight_scheduler(a, copies, dev, k): pass
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in frame_pairs.register_in_flight_scheduler(a, copies, dev, k): pass
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / len(copies) * 1e3)
    return sorted(ts)[2]
for k in (1, 2, 4, 8):
    print(f"in flight {k}: {stream(k):.3f} ms per frame pair")
orig = utils_match.sanity_grid
cache = {}
def cached(args, st, dt, si, di):
    key = (len(si), len(di))
    if key not in cache: cache[key] = orig(args, st, dt, si, di)
    return cache[key]
utils_match.sanity_grid = cached
print(f"in flight 4, sanity_grid from a cache: {stream(4):.3f} ms per frame pair")
