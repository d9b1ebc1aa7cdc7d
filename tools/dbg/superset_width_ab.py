import os, sys, time
ROOT="/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
ego = torch.eye(4, device=dev)
for mp in (2048, 10000):
    ref = None
    for w in (256, 512, 640, 768, 1024, 2048, 4096):
        a = frame_pairs.default_args(max_points=mp); a.device_association_width = w
        for _ in range(3): out = frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
        ts = []
        for _ in range(15):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        if w == 1024: ref = out["flow"].clone()
        print(f"max_points {mp} superset width {w}: median {sorted(ts)[7]:.3f} ms, pairs {len(out['pairs'])}", flush=True)
