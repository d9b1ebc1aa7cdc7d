"""Developer check: HDBSCAN / DBSCAN on a larger synthetic scene (several demo frames tiled side by side)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
base = np.concatenate([g["point_dst"], g["point_src"]], 0)
tiles = int(os.environ.get("TILES", "4"))
rng = np.random.default_rng(0)
pts = np.concatenate([base + np.array([250.0 * (k % 2), 250.0 * (k // 2), 0], np.float32) + rng.normal(0, 1e-3, base.shape).astype(np.float32)
                      for k in range(tiles)], 0)
tp = torch.from_numpy(pts).cuda()
print("points", len(pts))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = utils_cluster.hdbscan_mst(tp, 20)
    torch.cuda.synchronize(); print("spanning tree ms", (time.perf_counter() - t0) * 1e3)
a, b = t["a"].cpu().numpy(), t["b"].cpu().numpy()
nc, _ = connected_components(coo_matrix((np.ones(len(a)), (a, b)), shape=(len(pts), len(pts))), directed=False)
print("edges", len(a), "components", nc)
t0 = time.perf_counter(); lab = utils_cluster.hdbscan(tp, 20); print("hdbscan ms", (time.perf_counter() - t0) * 1e3, "clusters", lab.max() + 1)
t0 = time.perf_counter(); l2, sizes = utils_cluster.dbscan(tp, 0.25, 20); torch.cuda.synchronize(); print("dbscan ms", (time.perf_counter() - t0) * 1e3, "clusters", sizes.numel())
