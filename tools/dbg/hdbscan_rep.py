import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import utils_cluster
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
pts = torch.from_numpy(np.concatenate([g["point_dst"], g["point_src"]], 0)).cuda()
a = SimpleNamespace(min_cluster_size=20, num_clusters=200)
for _ in range(3): utils_cluster.cluster_hdbscan(a, pts)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): lab = utils_cluster.cluster_hdbscan(a, pts)
    torch.cuda.synchronize(); print("cluster_hdbscan %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
