"""Developer check: one batch of tools/dbg/registration_fuzz.py (SEED, TRIAL) in detail -- the ICP from the common
initial pose in the three HIP search modes, the fp32 oracle and the oracle with its Kabsch step evaluated in fp64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import _lib, synthetic, utils_hist, utils_helper, utils_icp_pytorch3d as hip_icp
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
target = int(os.environ.get("TRIAL", "6"))
for trial in range(target + 1):
    B = int(rng.integers(1, 24)); N = int(rng.choice([64, 128, 256, 384]))
    S, D, _ = synthetic.make_batch(B, N, seed=int(rng.integers(0, 10**6)), ragged=bool(rng.random() < 0.7), n_min=20)
    tf = float(rng.choice([2.0, 2.0, 3.34]))
a = rp.default_args(max_points=N, translation_frame=tf)
s, d = torch.from_numpy(S), torch.from_numpy(D)
n1, n2 = (s[:, :, 3] > 0).sum(1), (d[:, :, 3] > 0).sum(1)
sw = n1 > n2
A, C = s.clone(), d.clone(); A[sw] = d[sw]; C[sw] = s[sw]
init = rp.estimate_init_pose(a, A, C)
moved = rp.transform_points_batch(A, init)
o32 = rp.iterative_closest_point(moved, C)
o64 = rp.iterative_closest_point(moved, C, kabsch_dtype=torch.float64)
print("B", B, "N", N, "oracle iterations fp32 / fp64-kabsch:", o32.iterations, o64.iterations)
res = {}
for mode in ("scan", "grid", "sweep"):
    with _lib.options(search=mode):
        sol = hip_icp.iterative_closest_point(moved.to(dev), C.to(dev))
    res[mode] = sol
    print(mode, "iterations", sol.converged.iterations)
def disp(R1, T1, R2, T2):
    out = []
    for b in range(B):
        v = moved[b, :, 3] > 0
        p = moved[b, v, :3].double().numpy()
        out.append(np.abs((p @ np.asarray(R1[b], np.float64) + np.asarray(T1[b], np.float64)) - (p @ np.asarray(R2[b], np.float64) + np.asarray(T2[b], np.float64))).max())
    return np.array(out)
h = res["sweep"]
hR, hT = h.RTs.R.cpu().numpy(), h.RTs.T.cpu().numpy()
print("n_moving", (moved[:, :, 3] > 0).sum(1).tolist())
print("sweep vs oracle fp32   :", np.round(disp(hR, hT, o32.R.numpy(), o32.T.numpy()), 5).tolist())
print("sweep vs oracle kabsch64:", np.round(disp(hR, hT, o64.R.numpy(), o64.T.numpy()), 5).tolist())
print("oracle fp32 vs kabsch64 :", np.round(disp(o32.R.numpy(), o32.T.numpy(), o64.R.numpy(), o64.T.numpy()), 5).tolist())
print("sweep vs scan          :", np.round(disp(hR, hT, res["scan"].RTs.R.cpu().numpy(), res["scan"].RTs.T.cpu().numpy()), 7).tolist())
# inliers of the oracle's final state (gate d^2 <= thres^2): fewer than three leave the rotation undetermined
Xt = o64.Xt
for b in range(B):
    v = moved[b, :, 3] > 0; w = C[b, :, 3] > 0
    d2 = ((Xt[b, v, None, :3] - C[b, None, w, :3]) ** 2).sum(-1).min(1).values
    print("pair", b, "inliers at the end", int((d2 <= np.float32(0.1 * 0.1)).sum()), "of", int(v.sum()))
