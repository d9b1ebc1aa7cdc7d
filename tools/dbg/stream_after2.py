"""Developer tool: bench.extra_measurements, then the stream figure again (variant: VARIANT=noragged)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from types import SimpleNamespace
from conftest import load_golden
from icp_flow_amd import frame_pairs, synthetic, utils_match
dev = torch.device("cuda:0")
g, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
a10 = frame_pairs.default_args(max_points=10000)
def stream(tag):
    for _ in frame_pairs.register_in_flight(a10, [fp] * 4, dev, 4): pass
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in frame_pairs.register_in_flight(a10, [fp] * 12, dev, 4): pass
        torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t) / 12 * 1e3:.3f} ms / frame pair", flush=True)
args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50, icp_stop_mode="reference")
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
v = os.environ.get("VARIANT", "")
def step(): return utils_match.hist_icp(args, src, dst, return_iterations=True)
def sync(): torch.cuda.synchronize(dev)
if "steps50" in v: bench.timed_steps(step, sync, 50, 10, 50)
if "steps5_" in v: bench.timed_steps(step, sync, 5, 2, 50)
if "loop60" in v:
    for _ in range(60): step()
    sync()
T = utils_match.hist_icp(args, src, dst)
if "noragged" in v: bench.ragged_real_shape = lambda dev: {}
if "nocluster" in v:
    bench.cluster_measurement = lambda *a: {}
    bench.hdbscan_measurement = lambda *a: {}
e = bench.extra_measurements(args, src, dst, T, dev, SimpleNamespace())
print({m: (e["frame_pair"][m]["ms_per_frame_pair"], e["frame_pair"][m]["stream_ms_per_frame_pair_4_in_flight"]) for m in ("max_points_2048", "max_points_10000")})
stream("after extras " + v)
