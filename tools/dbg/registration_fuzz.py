"""Developer check: the fused registration against the oracle on many random ragged batches.  Initial poses must be
identical; the ICP from the common initial pose is compared on the pairs on which the oracle itself is stable (its
fp32 evaluation and the evaluation with an fp64 Kabsch step agree to 1e-5 m, at least four neighbours pass the gate in every iteration and three at the end: a flipping gate decision or an undetermined rotation is not a meaningful expectation); then the whole hist_icp on those pairs.
A pair may still differ by a millimetre: clusters sit tens of metres from the origin, so rotations that agree to 1e-7
move a point by micrometres, and a neighbour within that of the 0.1 m gate changes sides (all three HIP search modes
then agree with each other bit for bit: tools/dbg/registration_case.py).  Seen on about 1 pair in 200."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import synthetic, utils_match, utils_hist, utils_icp_pytorch3d as hip_icp
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
trials = int(os.environ.get("TRIALS", "30"))
worst_icp = worst_all = 0.0
init_diff = stable_pairs = all_pairs = iter_diff = 0


def disp(p, R1, T1, R2, T2):
    return float(np.abs((p @ np.asarray(R1, np.float64) + np.asarray(T1, np.float64)) -
                        (p @ np.asarray(R2, np.float64) + np.asarray(T2, np.float64))).max())


for trial in range(trials):
    B = int(rng.integers(1, 24)); N = int(rng.choice([64, 128, 256, 384]))
    S, D, _ = synthetic.make_batch(B, N, seed=int(rng.integers(0, 10**6)), ragged=bool(rng.random() < 0.7), n_min=20)
    tf = float(rng.choice([2.0, 2.0, 3.34]))
    a = rp.default_args(max_points=N, translation_frame=tf)
    s, d = torch.from_numpy(S), torch.from_numpy(D)
    sw = (s[:, :, 3] > 0).sum(1) > (d[:, :, 3] > 0).sum(1)
    A, C = s.clone(), d.clone(); A[sw] = d[sw]; C[sw] = s[sw]
    g0 = utils_hist.estimate_init_pose(a, A.to(dev), C.to(dev)).cpu()
    w0 = rp.estimate_init_pose(a, A, C)
    init_diff += int(((g0 - w0).abs().reshape(B, -1).max(1).values > 0).sum())
    moved = rp.transform_points_batch(A, w0)
    o32 = rp.iterative_closest_point(moved, C)
    o64 = rp.iterative_closest_point(moved, C, kabsch_dtype=torch.float64, trace=True)
    gated = torch.stack([h[3] for h in o64.history]).min(0).values   # fewest gated correspondences of any iteration
    h = hip_icp.iterative_closest_point(moved.to(dev), C.to(dev))
    iter_diff += int(h.converged.iterations != o64.iterations)
    hR, hT = h.RTs.R.cpu().numpy(), h.RTs.T.cpu().numpy()
    got = utils_match.hist_icp(a, s.to(dev), d.to(dev)).cpu().numpy()
    want = rp.hist_icp(a, s, d).numpy()
    for b in range(B):
        v = moved[b, :, 3] > 0
        p = moved[b, v, :3].double().numpy()
        all_pairs += 1
        if o32.iterations != o64.iterations or disp(p, o32.R[b].numpy(), o32.T[b].numpy(), o64.R[b].numpy(), o64.T[b].numpy()) > 1e-5:
            continue
        w = C[b, :, 3] > 0   # fewer than three gated neighbours at the end: the rotation is not determined (DESIGN 4.6 ii)
        d2 = ((o64.Xt[b, v, None, :3] - C[b, None, w, :3]) ** 2).sum(-1).min(1).values
        if int((d2 <= np.float32(0.1 * 0.1)).sum()) < 3 or int(gated[b]) < 4:   # (SEED=5 trial 28 pair 1: two in every iteration; SEED=6 trial 21
            # pair 7: three in iteration 0, sharing two targets -- a rank-1 covariance)
            continue
        stable_pairs += 1
        e = disp(p, hR[b], hT[b], o64.R[b].numpy(), o64.T[b].numpy())
        worst_icp = max(worst_icp, e)
        vs = S[b, :, 3] > 0
        ps = np.concatenate([S[b, vs, :3], np.ones((vs.sum(), 1), np.float32)], 1).astype(np.float64)
        e2 = float(np.abs(ps @ got[b].astype(np.float64).T - ps @ want[b].astype(np.float64).T).max())
        worst_all = max(worst_all, e2)
        if e > 1e-4 or e2 > 1e-4:
            print("above 1e-4 m (gate flip?)", trial, b, B, N, tf, e, e2)
print(f"trials {trials}: {all_pairs} pairs, {stable_pairs} stable in the oracle; different initial poses {init_diff}; "
      f"batches with another stop iteration {iter_diff}; worst ICP difference {worst_icp:.2e} m, worst hist_icp "
      f"difference {worst_all:.2e} m")
