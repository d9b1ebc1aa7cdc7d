"""Developer check: candidate scoring by pruned sorted sweeps (nn.hip launch_sweep_score_pruned) against the pruned
all-pairs scans and against every scan run to its end, on random batches: registrations must be bit-identical
(the pruning never changes the pick), and identical from run to run although which scans get pruned depends on timing."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
bad = pairs = 0
for seed in range(int(os.environ.get("FIRST", 0)), int(os.environ.get("FIRST", 0)) + int(os.environ.get("SEEDS", 40))):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([520, 600, 1024, 1500, 2048, 3000, 4096, 5000]))
    B = int(rng.integers(1, max(2, min(400, 400000 // N))))
    S, D, _ = synthetic.make_batch(B, N, seed=int(rng.integers(0, 10**6)), ragged=bool(rng.random() < 0.7), n_min=int(rng.choice([5, 40, 300])))
    if rng.random() < 0.3:   # far from the origin: the windows' rounding slack matters
        off = rng.uniform(-900, 900, (B, 1, 3)).astype(np.float32)
        S[:, :, :3] += off * (S[:, :, 3:4] > 0); D[:, :, :3] += off * (D[:, :, 3:4] > 0)
    tf = float(rng.choice([2.0, 3.34, 10.02]))
    a = rp.default_args(max_points=N, translation_frame=tf, icp_max_iterations=int(rng.choice([5, 50])))
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    T1 = utils_match.hist_icp(a, s, d)
    with _lib.options(no_score_sweep=True):
        T0 = utils_match.hist_icp(a, s, d)
    with _lib.options(no_score_prune=True):
        T2 = utils_match.hist_icp(a, s, d)
    again = all(torch.equal(utils_match.hist_icp(a, s, d), T1) for _ in range(2))
    same = lambda A, Bt: bool(((A == Bt) | (torch.isnan(A) & torch.isnan(Bt))).all())
    ok = same(T0, T1) and same(T2, T1) and again
    pairs += B
    if not ok:
        bad += 1
        diff = (~((T0 == T1) | (torch.isnan(T0) & torch.isnan(T1)))).flatten(1).any(1).sum().item()
        print(f"seed {seed}: B {B} N {N} tf {tf}: DIFFERENT (pairs differing from the pruned scans: {diff}, run to run equal: {again})")
print(f"{pairs} pairs in {int(os.environ.get('SEEDS', 40))} batches: {bad} batches differ")
