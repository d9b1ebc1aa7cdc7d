"""Developer tool: the largest cluster pair of the demo frame (a 29 m wall, 10 000 sampled points) alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_hist, utils_match, frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
N = int(os.environ.get("MP", "10000")); B = int(os.environ.get("B", "1"))
rng = np.random.default_rng(0)
def seg(p, l):
    a = p[l == 145.0]; a = a[rng.permutation(len(a))[:N]]
    out = np.full((N, 4), 1e8, np.float32); out[:, 3] = 0; out[:len(a), :3] = a; out[:len(a), 3] = 1
    return out
S = torch.from_numpy(np.stack([seg(g["point_src"], lab["label_src"])] * B)).cuda()
D = torch.from_numpy(np.stack([seg(g["point_dst"], lab["label_dst"])] * B)).cuda()
a = frame_pairs.default_args(max_points=N)
for name, fn in (("estimate_init_pose", lambda: utils_hist.estimate_init_pose(a, S, D)), ("hist_icp", lambda: utils_match.hist_icp(a, S, D))):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 5 * 1e3, "ms")
