"""Developer tool: cProfile of the host thread over demo frame pairs registered ONE AT A TIME (the latency figure): what sits
between the kernels."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs, utils_flow, utils_track
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=10000); a.native_host = False   # (a profile of the Python host)
eye = torch.eye(4, device=dev)
def run():
    torch.manual_seed(0)
    pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
    return utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, eye)
for _ in range(3): run()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): run()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps_ = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps_.print_stats(32)
print(s.getvalue()[:7000])
