"""Developer tool: the demo frame pair through icpflow_track_frame, one at a time, at max_points 2048 and 10000 (inputs resident),
with the library selected by ICPFLOW_HIP_LIB."""
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
out = []
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    for _ in range(3): frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize(); t = time.perf_counter()
        frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    out.append(f"max_points {mp}: median {sorted(ts)[7]:.3f} ms")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), " | ".join(out))
