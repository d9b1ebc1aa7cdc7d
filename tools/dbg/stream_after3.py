"""Developer tool: what makes the first in-flight stream at max_points 10000 slow after many hist_icp calls? VARIANT=..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from types import SimpleNamespace
from conftest import load_golden
from icp_flow_amd import frame_pairs, synthetic, utils_match, _lib
dev = torch.device("cuda:0")
v = os.environ.get("VARIANT", "")
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
def stream(tag, mp):
    a10 = frame_pairs.default_args(max_points=mp)
    for _ in frame_pairs.register_in_flight(a10, [fp] * 4, dev, 4): pass
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in frame_pairs.register_in_flight(a10, [fp] * 12, dev, 4): pass
        torch.cuda.synchronize(); print(f"{v} {tag} mp {mp}: {(time.perf_counter() - t) / 12 * 1e3:.3f} ms / frame pair", flush=True)
args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50, icp_stop_mode="reference")
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
if "prestream" in v:
    pool = frame_pairs._stream_pool.setdefault((dev.type, dev.index), [])
    while len(pool) < 4: pool.append(torch.cuda.Stream(dev))
n = 60 if "n60" in v else 7
if "prof" in v:
    prof = _lib.Profile(n + 8)
    with _lib.options(profile=prof):
        for _ in range(n): utils_match.hist_icp(args, src, dst, return_iterations=True)
    torch.cuda.synchronize(); prof.collect(); prof.close()
else:
    for _ in range(n): utils_match.hist_icp(args, src, dst, return_iterations=True)
    torch.cuda.synchronize()
if "sleep" in v: time.sleep(2.0)
if "first2048" in v: stream("first", 2048)
stream("then", 10000)
