"""Developer tool: cProfile of the host thread while the demo frame pair runs as a stream with frame pairs in flight."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000")))
K = int(os.environ.get("K", "4"))
for _ in frame_pairs.register_in_flight(a, [fp] * 8, dev, K): pass
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
n = 0
for _ in frame_pairs.register_in_flight(a, [fp] * 24, dev, K): n += 1
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("ROWS", "40")))
