"""Developer tool (library built with -DICPFLOW_DEBUG_EXECUTED): executed ICP iterations per pair in the
first association stage of the demo frame (max_points from MP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs, utils_hist, utils_helper, utils_icp_pytorch3d as icp
from icp_flow_amd.utils_check import ClusterTable, _sanity_mask
from icp_flow_amd.utils_match import _gather_pair_batches
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000")))
torch.manual_seed(0)
st, dt = ClusterTable.pair(G(g["point_src"]), G(lab["label_src"]).float(), G(g["point_dst"]), G(lab["label_dst"]).float())
lu = np.unique(np.concatenate([st.h_labels.astype(np.int64), dt.h_labels.astype(np.int64)]))
pairs = np.stack([lu, lu], 1); pairs = pairs[pairs.min(1) >= 0].astype(np.float32)
pairs = pairs[_sanity_mask(a, st, dt, pairs)]
si, di = st.find_host(pairs[:, 0]), dt.find_host(pairs[:, 1])
S, D = _gather_pair_batches(a, st, dt, si, di)
# same roles as hist_icp: smaller cloud is the moving one
ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
sw = ns > nd
A = torch.where(sw[:, None, None], D, S); C = torch.where(sw[:, None, None], S, D)
init = utils_hist.estimate_init_pose(a, A, C)
sol = icp.iterative_closest_point(utils_helper.transform_points_batch(A, init), C, max_iterations=100)
ex = sol.rmse.cpu().numpy()
print("pairs", len(pairs), "stop iteration", sol.converged.iterations)
order = np.argsort(-ex)
for k in order[:12]:
    print("pair %3d  n_moving %5d  n_fixed %5d  executed %3d" % (k, int(torch.minimum(ns, nd)[k]), int(torch.maximum(ns, nd)[k]), int(ex[k])))
