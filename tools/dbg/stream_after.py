"""Developer tool: the stream-of-frame-pairs figure at max_points 10000 before / after the other bench extras (what slows it?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from conftest import load_golden
from icp_flow_amd import frame_pairs, synthetic, utils_match
dev = torch.device("cuda:0")
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
a = frame_pairs.default_args(max_points=10000)
def stream(tag):
    for _ in frame_pairs.register_in_flight(a, [fp] * 4, dev, 4): pass
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in frame_pairs.register_in_flight(a, [fp] * 12, dev, 4): pass
        torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t) / 12 * 1e3:.3f} ms / frame pair", flush=True)
stream("fresh process")
from types import SimpleNamespace
from icp_flow_amd import _lib, utils_icp, utils_helper, utils_hist, hist as hip_hist
args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50, icp_stop_mode="reference")
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
def step():
    return utils_match.hist_icp(args, src, dst, return_iterations=True)
def sync(): torch.cuda.synchronize(dev)
dt, icp_ms, icp_launches, T, iters = bench.timed_steps(step, sync, 50, 10, 50)
stream("after timed_steps")
utils_match.match_eval(args, src, dst, T); sync()
stream("after match_eval")
with _lib.options(search="scan"):
    utils_match.hist_icp(args, src, dst); sync()
stream("after scan search")
fast = SimpleNamespace(**{**vars(args), "icp_stop_mode": "per_pair"})
utils_match.hist_icp(fast, src, dst); sync()
stream("after per-pair")
init = torch.eye(4, device=dev)[None].repeat(256, 1, 1).contiguous()
utils_icp.apply_icp(args, src, dst, init); sync()
stream("after apply_icp")
ex, ey, ez = utils_hist.bin_edges(args)
hip_hist.hist(dst, src, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez)); sync()
stream("after hist")
utils_helper.nearest_neighbor_batch(src, dst); sync()
stream("after nn")
