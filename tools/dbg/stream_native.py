"""Developer tool: the stream of demo frame pairs through icpflow_track_frame with one host thread per frame pair in flight
(frame_pairs.register_in_flight_native) against the generator-based scheduler on one host thread (register_in_flight_scheduler)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
n = int(os.environ.get("FRAMES", "48"))
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    a.teams_full_gpu = os.environ.get("TEAMS_FULL_GPU") == "1"
    a.device_association = True
    ref = frame_pairs.register_frame_pair(a, fp, dev)["flow"]
    a.device_association = None
    for _ in range(3): frame_pairs.register_frame_pair_native(a, fp, dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): frame_pairs.register_frame_pair_native(a, fp, dev)
    torch.cuda.synchronize(); print(f"max_points {mp}: native, one at a time {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per frame pair")
    for _ in range(3): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): frame_pairs.register_frame_pair(a, fp, dev)
    torch.cuda.synchronize(); print(f"max_points {mp}: Python host, one at a time {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per frame pair")
    for k in (2, 3, 4, 6, 8):
        for fn, name in ((frame_pairs.register_in_flight_native, "native threads"), (frame_pairs.register_in_flight_scheduler, "one-thread scheduler")):
            for _ in fn(a, [fp] * (2 * k), dev, k): pass
            torch.cuda.synchronize(); t = time.perf_counter()
            flows = [o["flow"] for _, _, o in fn(a, [fp] * n, dev, k)]
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            bad = sum(0 if torch.equal(f, ref) else 1 for f in flows) if fn is frame_pairs.register_in_flight_native else 0
            del flows
            print(f"max_points {mp}: {k} in flight, {name}: {dt / n * 1e3:.3f} ms per frame pair" + (f", flows different from the one-at-a-time device path: {bad}" if fn is frame_pairs.register_in_flight_native else ""), flush=True)
