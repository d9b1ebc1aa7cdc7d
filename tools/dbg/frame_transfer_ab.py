import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_flow, utils_match, frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
eye = torch.eye(4, device=dev)
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    for mode in (False, True, False, True):
        def run():
            torch.manual_seed(0)
            pairs, Tm = utils_match.drive(utils_match.match_pcds_steps(a, ps, pd, ls, ld, asynchronous=mode))
            return utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, eye)
        for _ in range(3): run()
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t = time.perf_counter(); f = run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(mp, "pinned+event" if mode else "pageable", "median %.3f ms" % sorted(ts)[4], "min %.3f" % min(ts))
