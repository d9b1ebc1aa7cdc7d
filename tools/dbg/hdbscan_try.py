"""Developer check: icpflow_hdbscan_mst vs the oracle tree on crops of the demo frame, then timing on the
whole frame pair."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster
from oracle import hdbscan as oh
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
pts = np.concatenate([g["point_dst"], g["point_src"]], 0)
for (cx, cy, n, k) in [(0, 5, 1500, 20), (20, -10, 3000, 20), (-15, 20, 2500, 30), (0, 0, 700, 5)]:
    sel = (np.abs(pts[:, 0] - cx) < 15) & (np.abs(pts[:, 1] - cy) < 12)
    P = pts[sel][:n]
    t = utils_cluster.hdbscan_mst(P, k)
    a, b, w2 = t["a"].cpu().numpy().astype(np.int64), t["b"].cpu().numpy().astype(np.int64), t["w2"].cpu().numpy()
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    o = np.lexsort((hi, lo)); lo, hi, w2 = lo[o], hi[o], w2[o]
    ra, rb, rw, rc = oh.mst(P, k)
    print(len(P), "edges", len(lo), "core equal", np.array_equal(t["core2"].cpu().numpy(), rc),
          "edges equal", np.array_equal(lo, ra) and np.array_equal(hi, rb), "weights equal", np.array_equal(w2, rw),
          "sum", w2.sum(), rw.sum())
tp = torch.from_numpy(pts).cuda()
for _ in range(2):
    t = utils_cluster.hdbscan_mst(tp, 20)
torch.cuda.synchronize(); t0 = time.perf_counter()
t = utils_cluster.hdbscan_mst(tp, 20)
torch.cuda.synchronize(); print("full frame mst ms", (time.perf_counter() - t0) * 1e3, "edges", len(t["a"]))
t0 = time.perf_counter()
lab = utils_cluster.hdbscan(tp, 20)
print("full hdbscan ms", (time.perf_counter() - t0) * 1e3, "clusters", lab.max() + 1, "noise", (lab == -1).sum())
L = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo_labels.npz"))
ref = np.concatenate([L["label_dst"], L["label_src"]]).astype(np.int64)
from types import SimpleNamespace
a = SimpleNamespace(min_cluster_size=20, num_clusters=200, if_hdbscan=True)
t0 = time.perf_counter()
mine = utils_cluster.cluster_pcd(a, pts, np.ones(len(pts), bool)).astype(np.int64)
print("cluster_pcd(hdbscan) ms", (time.perf_counter() - t0) * 1e3)
print("labels differ at", int((mine != ref).sum()), "of", len(ref), "points; clusters", mine.max() + 1, ref.max() + 1)
from sklearn.metrics import adjusted_rand_score
print("ARI", adjusted_rand_score(ref, mine))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); utils_cluster.hdbscan(tp, 20); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
