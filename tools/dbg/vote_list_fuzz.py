"""Developer tool: the vote's work list (hist.hip: vote_plan_kernel) and count_pair's boxes against the grid over the padded widths,
on random ragged batches of long clouds -- bins and registrations compared bit for bit (ICPFLOW_OPT_NO_VOTE_LIST), the bins of the
small trials also against the oracle's vote.  TRIALS=40 python tools/dbg/vote_list_fuzz.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
bad = 0
for trial in range(int(os.environ.get("TRIALS", 30))):
    B = int(rng.choice([1, 2, 3, 7, 16, 40, 130, 300]))
    N = int(rng.integers(4097, 4097 + (8000 if B <= 40 else 600)))
    S, D, _ = synthetic.make_batch(B, N, seed=int(rng.integers(1 << 30)), ragged=True, n_min=int(rng.choice([1, 20, 300])))
    for side in (S, D):                       # a few pairs without rows, a few with every row
        for b in rng.choice(B, size=max(1, B // 8), replace=False):
            if rng.random() < 0.5:
                side[b, :, 3] = 0.0; side[b, :, :3] = 1e8
    a = rp.default_args(max_points=N, icp_max_iterations=int(rng.choice([3, 30])))
    ex, ey, ez = rp.bin_edges(a)
    L = len(ex) * len(ey) * len(ez)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    out = []
    for plain in (False, True):
        bins = torch.full((B, L), -1, dtype=torch.int32, device=dev)
        with _lib.options(vote_bins=bins, no_vote_list=plain):
            T = utils_match.hist_icp(a, s, d)
        out.append((bins.cpu().numpy(), T.cpu().numpy()))
    same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1], equal_nan=True)
    note = ""
    if B * N <= 3 * 9000:                     # the oracle's all-pairs vote: seconds
        n1, n2 = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
        sw = n1 > n2
        src, dst = torch.from_numpy(S).clone(), torch.from_numpy(D).clone()
        src[sw], dst[sw] = torch.from_numpy(D)[sw], torch.from_numpy(S)[sw]
        want = rp.hist(dst, src, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez)).numpy().reshape(B, L)
        ok = np.array_equal(out[0][0].view(np.uint32).astype(np.int64), want.astype(np.int64))
        note = " oracle bins " + ("equal" if ok else "DIFFER")
        same = same and ok
    bad += 0 if same else 1
    print(f"trial {trial}: {B} x {N}: {'same' if same else 'DIFFERENT'}{note}", flush=True)
print("different:", bad)
