"""Developer tool: many passes of the demo frame-pair stream with 2 ... 8 frame pairs in flight (two team launches side by side,
ICPFLOW_OPT_TEAMS_HALF_GPU): every flow must equal the one-at-a-time flow bit for bit, no team may time out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
fps = []
for tag in ("g8_demo", "g8_demo_mp10000"):
    g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
    fps.append(frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"]))
bad = 0
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    a.device_association = {"1": True, "0": False}.get(os.environ.get("DEVICE_ASSOC", ""), None)
    a.native_host = os.environ.get("NATIVE", "1") == "1"   # (0: the generator-based scheduler on one host thread)
    ref = [o["flow"] for _, _, o in frame_pairs.register_in_flight(a, fps[:1], dev, 1)][0]
    for k in (2, 3, 4, 8):
        t = time.perf_counter(); n = 0
        for rep in range(int(os.environ.get("REPS", "12"))):
            for i, _, o in frame_pairs.register_in_flight(a, [fps[0]] * 12, dev, k):
                n += 1
                if not torch.equal(o["flow"], ref): bad += 1
        torch.cuda.synchronize()
        print(f"native host {os.environ.get('NATIVE', '1')}, device association {os.environ.get('DEVICE_ASSOC', 'default')}, max_points {mp}, {k} in flight: {n} frame pairs, {(time.perf_counter() - t) / n * 1e3:.3f} ms each, different flows so far {bad}")
print("different:", bad)
