"""Developer tool: the stream of demo frame pairs (native host, 4 and 8 in flight, max_points 2048 / 10000), median of five
passes of 32 frame pairs, with the library selected by ICPFLOW_HIP_LIB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
out = []
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp); a.device_association_width = int(os.environ.get("WIDTH", "1024"))
    for k in (4, 8):
        for _ in frame_pairs.register_in_flight(a, [fp] * (2 * k), dev, k): pass
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in frame_pairs.register_in_flight(a, [fp] * 32, dev, k): pass
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 32 * 1e3)
        out.append(f"{mp}/{k} in flight {sorted(ts)[2]:.3f}")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), "stream ms per frame pair:", " | ".join(out))
