"""Developer tool: the demo frame pair as a stream with 4 in flight -- inputs uploaded per frame pair against resident inputs
(frame_pairs.make_resident), 12 / 32 / 64 copies, several passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000")))
for resident in (False, True, False, True):
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
    if resident: frame_pairs.make_resident(fp, dev)
    for copies in (12, 32, 64):
        fps = [fp] * copies
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in frame_pairs.register_in_flight(a, fps, dev, 4): pass
            torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) / copies * 1e3, 3))
        print("resident" if resident else "uploaded", copies, ts)
