import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
copies = [fp] * 12
for mp in (2048, 10000):
    for mode in (False, True, False, True):
        a = frame_pairs.default_args(max_points=mp); a.device_association = mode
        def stream(k):
            for _ in frame_pairs.register_in_flight_scheduler(a, copies, dev, k): pass
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in frame_pairs.register_in_flight_scheduler(a, copies, dev, k): pass
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / len(copies) * 1e3)
            return sorted(ts)[2]
        print(mp, "device" if mode else "host  ", " ".join(f"{k} in flight {stream(k):.3f}" for k in (1, 2, 4, 8)))
