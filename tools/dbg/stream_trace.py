"""Developer tool: a short stream of demo frame pairs (HOST=native|scheduler, MODE=device|host association, K in flight, MP =
max_points) for rocprofv3 --kernel-trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000"))); a.device_association = os.environ.get("MODE", "device") == "device"
run = frame_pairs.register_in_flight_native if os.environ.get("HOST", "scheduler") == "native" else frame_pairs.register_in_flight_scheduler
k = int(os.environ.get("K", "4"))
for _ in run(a, [fp] * 8, dev, k): pass
torch.cuda.synchronize(); t = time.perf_counter()
for _ in run(a, [fp] * 16, dev, k): pass
torch.cuda.synchronize(); print("ms per frame pair", (time.perf_counter() - t) / 16 * 1e3)
