"""Developer tool: cProfile of the host thread over a stream of demo frame pairs (4 in flight): where its time goes."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
a = frame_pairs.default_args(max_points=10000); a.device_association = os.environ.get("DEVICE_ASSOC", "1") == "1"
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
copies = [fp] * 24
for _ in frame_pairs.register_in_flight_scheduler(a, copies[:8], dev, 4): pass
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in frame_pairs.register_in_flight_scheduler(a, copies, dev, 4): pass
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("cumulative"); ps.print_stats(45)
print(s.getvalue()[:9000])
