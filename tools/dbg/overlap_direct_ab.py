"""Developer tool: the demo frame pair one at a time through track_frame_native (direct call, default stream), stage 2's initial
poses beside stage 1's ICP or behind it, alternating in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
ego = torch.eye(4, device=dev)
side = torch.cuda.Stream(dev)
for where in ("default stream", "a torch stream"):
    for mp in (2048, 10000):
        for ov in (False, True, False, True):
            a = frame_pairs.default_args(max_points=mp); a.stage_overlap = ov
            def run():
                if where == "default stream":
                    frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
                else:
                    with torch.cuda.stream(side):
                        frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
            for _ in range(3): run()
            ts = []
            for _ in range(15):
                torch.cuda.synchronize(); t = time.perf_counter()
                run()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            print(f"{where}, max_points {mp}, overlap {ov}: median {sorted(ts)[7]:.3f} ms")
