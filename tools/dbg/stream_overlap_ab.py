"""Developer tool: the demo frame pair as a stream with 4 in flight and one at a time, stage 2's initial poses beside stage 1's ICP
(args.stage_overlap = True) or behind it (False)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
fp = frame_pairs.make_resident(frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"]), dev)
for mp in (2048, 10000):
    for ov in (False, True, None, False, True):
        a = frame_pairs.default_args(max_points=mp); a.stage_overlap = ov
        for infl in (4, 1):
            fps = [fp] * 48
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in frame_pairs.register_in_flight(a, fps, dev, infl): pass
                torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) / len(fps) * 1e3, 3))
            print(f"max_points {mp} overlap {ov} in flight {infl}: {ts}")
