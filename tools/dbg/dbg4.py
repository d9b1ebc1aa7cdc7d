import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_icp, utils_helper
from oracle import reference_path as rp
g = load_golden("g6_hist_icp")
a = rp.default_args(translation_frame=float(g["translation_frame"]))
src, dst, init = torch.from_numpy(g["src"]), torch.from_numpy(g["dst"]), torch.from_numpy(g["T_init_noswap"])
M, aux = rp.apply_icp(a, src, dst, init.clone(), return_aux=True)
print("oracle iters", aux["iterations"], "e0", aux["error_init"].numpy(), "e1", aux["error_icp"].numpy(), aux["rolled_back"].numpy())
got, it = utils_icp.apply_icp(a, src.cuda(), dst.cuda(), init.cuda(), return_iterations=True)
print("gpu iters", int(it))
got = got.cpu()
print("pair1 got\n", got[1].numpy(), "\nwant\n", M[1].numpy(), "\ninit\n", init[1].numpy())
# gpu-side errors via nn
for name, P in (("init", init), ("final", got)):
    mv = utils_helper.transform_points_batch(src.cuda(), P.cuda())
    _, d = utils_helper.nearest_neighbor_batch(mv, dst.cuda())
    m = (src[:,:,3]>0).cuda()
    print(name, ((d*m).sum(1)/m.sum(1)).cpu().numpy())
