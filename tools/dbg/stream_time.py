"""Developer tool: the demo frame pair as a stream: one at a time against 2 / 4 / 8 frame pairs in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    n = 16
    for _ in range(2): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); print(f"max_points {mp}: one at a time (host upload included) {(time.perf_counter() - t) / n * 1e3:.3f} ms / frame pair")
    for k in (2, 3, 4, 6, 8):
        for _ in frame_pairs.register_in_flight_scheduler(a, [fp] * k, dev, k): pass
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in frame_pairs.register_in_flight_scheduler(a, [fp] * n, dev, k): pass
        torch.cuda.synchronize(); print(f"   {k} in flight: {(time.perf_counter() - t) / n * 1e3:.3f} ms / frame pair")
