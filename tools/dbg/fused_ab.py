"""Developer tool: hist_icp + match_eval as two calls against icpflow_hist_icp_eval; the demo frame pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match, frame_pairs
from conftest import load_golden
dev = torch.device("cuda", 0)
for B, N, reps in ((256, 1024, 40), (1024, 2048, 8), (8192, 2048, 3)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    def two():
        T = utils_match.hist_icp(a, s, d); return utils_match.match_eval(a, s, d, T)
    def one():
        return utils_match.hist_icp_eval(a, s, d)
    def plain():
        return utils_match.hist_icp(a, s, d)
    res = []
    for fn in (plain, two, one):
        best = 1e9
        for rep in range(3):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / reps * 1e3)
        res.append(best)
    print(f"{B}x{N}: hist_icp {res[0]:.4f} ms, + match_eval (two calls) {res[1]:.4f}, hist_icp_eval {res[2]:.4f}", flush=True)
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    for _ in range(3): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(16): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); print(f"demo frame pair, max_points {mp}: {(time.perf_counter() - t) / 16 * 1e3:.3f} ms (upload included)")
