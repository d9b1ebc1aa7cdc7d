"""Developer tool (library built with -DICPFLOW_DEBUG_EXECUTED): how many ICP iterations each pair of
BASELINE config 2 actually executes before its trajectory is periodic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_hist, utils_icp_pytorch3d as icp, utils_helper
B, N = 256, 1024
S, D, _ = synthetic.make_batch(B, N, seed=0)
src, dst = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N)
init = utils_hist.estimate_init_pose(a, src, dst)
moved = utils_helper.transform_points_batch(src, init)
sol = icp.iterative_closest_point(moved, dst, max_iterations=50)
ex = sol.rmse.cpu().numpy()
print("stop iteration", sol.converged.iterations, "executed: mean %.1f median %.0f max %.0f" % (ex.mean(), np.median(ex), ex.max()))
print("pairs still running after k iterations:", {k: int((ex > k).sum()) for k in (8, 12, 16, 20, 24, 32, 40)})
