import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_icp_pytorch3d as icp, utils_helper
from oracle import reference_path as rp
g = load_golden("g6_hist_icp")
src, dst, init = torch.from_numpy(g["src"]), torch.from_numpy(g["dst"]), torch.from_numpy(g["T_init_noswap"])
moved = rp.transform_points_batch(src, init)
mg = utils_helper.transform_points_batch(src.cuda(), init.cuda())
print("moved diff", float((mg.cpu()-moved)[src[:,:,3]>0].abs().max()))
sol = rp.iterative_closest_point(moved, dst, trace=True)
b = 1
for k in range(1, sol.iterations+1):
    s = icp.iterative_closest_point(moved.cuda(), dst.cuda(), max_iterations=k)
    R = s.RTs.R.cpu().numpy(); T = s.RTs.T.cpu().numpy()
    hR, hT, hr, hw = sol.history[k-1]
    print(k, "dR", np.abs(R[b]-hR[b].numpy()).max(), "dT", np.abs(T[b]-hT[b].numpy()).max(), "rmse", float(s.rmse[b]), float(hr[b]), "w", int(hw[b]))
    # inlier set on GPU side: recompute with oracle knn from GPU's previous R,T
