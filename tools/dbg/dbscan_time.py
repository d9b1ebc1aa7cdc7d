"""Time icpflow_dbscan on the demo frame pair (126 598 points) and on its halves."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
pts = torch.from_numpy(np.concatenate([g["point_dst"], g["point_src"]], 0)).cuda()
for eps, mp in ((0.25, 20), (0.25, 30), (0.4, 10)):
    for _ in range(3):
        lab, sizes = utils_cluster.dbscan(pts, eps, mp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        lab, sizes = utils_cluster.dbscan(pts, eps, mp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"dbscan n={len(pts)} eps={eps} min_points={mp}: {dt:.3f} ms  clusters {sizes.numel()}")
