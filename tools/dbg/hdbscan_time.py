"""Time icpflow_hdbscan_mst + the host remainder on the demo frame pair (126 598 points)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
pts = torch.from_numpy(np.concatenate([g["point_dst"], g["point_src"]], 0)).cuda()
for k in (20, 30):
    for _ in range(2):
        t = utils_cluster.hdbscan_mst(pts, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        t = utils_cluster.hdbscan_mst(pts, k)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    a, b, w = t["a"].cpu().numpy(), t["b"].cpu().numpy(), np.sqrt(t["w2"].cpu().numpy())
    t0 = time.perf_counter()
    lab = utils_cluster.labels_from_mst(a, b, w, len(pts), k)
    t1 = time.perf_counter()
    o = np.argsort(w, kind="stable")
    a2, b2, w2 = a[o].copy(), b[o].copy(), w[o].copy()
    t2 = time.perf_counter()
    lab2 = utils_cluster.labels_from_mst(a2, b2, w2, len(pts), k)
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    for _ in range(5):
        full = utils_cluster.hdbscan(pts, k - 1)
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"min_samples {k}: spanning tree {ms:.2f} ms, host remainder {(t1 - t0) * 1e3:.2f} ms (edges as they come), "
          f"{(t3 - t2) * 1e3:.2f} ms (edges sorted by weight), same labels {np.array_equal(lab, lab2)}, "
          f"whole hdbscan() {(t5 - t4) / 5 * 1e3:.2f} ms, {lab.max() + 1} clusters")
