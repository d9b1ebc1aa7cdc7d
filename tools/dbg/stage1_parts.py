"""Developer tool: stage 1 of the demo frame pair by subsets of its pairs -- which pairs pace the vote, the scoring, the ICP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, utils_track, utils_match, utils_hist, frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000"))); a.native_host = False   # (the hook below sits in the Python host)
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
S, D, cs, cd = kept[0]
B = len(cs)
big = np.minimum(cs, cd)
order = np.argsort(-big)
print("stage 1:", B, "pairs, width", S.shape[1], "; the largest (src x dst):", [(int(cs[k]), int(cd[k])) for k in order[:6]])
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
def report(name, idx):
    idx = torch.as_tensor(np.asarray(idx), device=dev)
    s, d = S[idx].contiguous(), D[idx].contiguous()
    prof = _lib.Profile(64)
    t_init = timed(lambda: utils_hist.estimate_init_pose_batch(a, s, d))
    with _lib.options(profile=prof):
        t_all = timed(lambda: utils_match.hist_icp_eval(a, s, d))
    icp, n = prof.collect()
    print(f"{name:34s} {len(idx):3d} pairs: estimate_init_pose {t_init:.3f} ms, hist_icp_eval {t_all:.3f} ms (ICP launch {icp / n:.3f})")
report("all", np.arange(B))
report("the largest pair alone", order[:1])
report("all but the largest", order[1:])
report("all but the six largest", order[6:])
report("the six largest", order[:6])
report("the second largest alone", order[1:2])
