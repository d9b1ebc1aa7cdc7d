"""Developer tool: the trajectories of stage 1 of the demo frame pair (max_points MP, default 10000): for every pair the first
iteration at which its state (R, T) equals, bit for bit, an EARLIER state of its own, and the distance back (the period) --
what a longer memory than the kernel's eight states would catch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, utils_track, utils_match, utils_hist, utils_helper, frame_pairs
from icp_flow_amd.utils_icp_pytorch3d import iterative_closest_point
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000"))); a.native_host = False
kept = []
orig = utils_match._register_stage
def stash(args, st, dt, si, di, *rest):
    out = orig(args, st, dt, si, di, *rest)
    stage, scratch = out[1], out[2][3]
    clouds = scratch[: 2 * stage.K * stage.N * 4].view(2, stage.K, stage.N, 4)
    kept.append((clouds[0].clone(), clouds[1].clone(), st.h_count[si].copy(), dt.h_count[di].copy()))
    return out
utils_match._register_stage = stash
torch.manual_seed(0)
utils_track.track(a, ps, pd, ls, ld)
utils_match._register_stage = orig
for stage, (S, D, cs, cd) in enumerate(kept[:2]):
    n1, n2 = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    sw = n1 > n2
    A, B = S.clone(), D.clone()
    A[sw], B[sw] = D[sw], S[sw]
    init = utils_hist.estimate_init_pose(a, A, B)
    X0 = utils_helper.transform_points_batch(A, init)
    sol = iterative_closest_point(X0, B, thres=a.thres_dist, max_iterations=100, relative_rmse_thr=1e-6)
    rec = sol.t_history.records().cpu().numpy()            # [it, B, 16]
    it, K = rec.shape[0], rec.shape[1]
    print(f"stage {stage + 1}: {K} pairs, {it} iterations (batch rule)")
    state = rec[:, :, :12].view(np.uint32)
    rows = []
    for b in range(K):
        seen = {}
        first = None
        for k in range(it):
            key = state[k, b].tobytes()
            if key in seen:
                first = (k, k - seen[key]); break
            seen[key] = k
        rows.append((b, int(min(n1[b], n2[b])), int(max(n1[b], n2[b])), first))
    rows.sort(key=lambda r: -(r[3][0] if r[3] else 10**6))
    for b, ns, nd, first in rows[:12]:
        print(f"   pair {b:3d}: {ns:5d} x {nd:5d} points: " + (f"state of iteration {first[0]} = state {first[1]} iterations earlier" if first else "no state repeats within the run"))
    per = [r[3][1] for r in rows if r[3]]
    print("   periods:", np.bincount(per).tolist(), "| pairs without a repeat:", sum(1 for r in rows if not r[3]))
