import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_icp_pytorch3d as icp
from oracle import reference_path as rp
g = load_golden("g5_icp")
X, Y = g["e_src"], g["e_dst"]
print("n", (X[:,:,3]>0).sum(1), (Y[:,:,3]>0).sum(1), "ref iters", g["e_iterations"])
sol = rp.iterative_closest_point(torch.from_numpy(X), torch.from_numpy(Y), trace=True)
for k in range(1, int(g["e_iterations"])+1):
    s = icp.iterative_closest_point(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), max_iterations=k)
    R = s.RTs.R.cpu().numpy(); T = s.RTs.T.cpu().numpy()
    dR = np.abs(R - g["e_hist_R"][k-1]).max((1,2)); dT = np.abs(T - g["e_hist_T"][k-1]).max(1)
    print(k, "iters", s.converged.iterations, "dR", np.array2string(dR, precision=2), "dT", np.array2string(dT, precision=2),
          "rmse", np.array2string(s.rmse.cpu().numpy(), precision=6), "ref rmse", np.array2string(sol.history[k-1][2].numpy(), precision=6), "w", sol.history[k-1][3].numpy())
