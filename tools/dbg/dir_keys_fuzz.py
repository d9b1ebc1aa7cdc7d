"""Developer tool: the sweeps' direction sort keys (csrc/sortdir.hpp) against axis keys (ICPFLOW_OPT_NO_DIR_KEYS) on many random batches,
some far from the origin: same initial poses and iteration counts, transforms equal to the order of the fp64 sums (a point moves by less
than an ulp of its coordinates), match_eval metrics equal to rounding."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_hist, utils_match
from oracle import reference_path as rp
bad = nbits = npairs = 0
worst = 0.0
for seed in range(int(os.environ.get("FIRST", 0)), int(os.environ.get("FIRST", 0)) + int(os.environ.get("SEEDS", 40))):
    rng = np.random.default_rng(seed)
    B, N = int(rng.integers(8, 300)), int(rng.choice([1100, 1500, 2048, 3000, 4096]))
    if seed % 5 == 4:
        B, N = (int(rng.integers(4, 40)), int(rng.choice([5000, 6000, 10000]))) if seed % 2 else (int(rng.integers(512, 1300)), int(rng.choice([1500, 2048])))
    S, D, _ = synthetic.make_batch(B, N, seed=1000 * seed, ragged=("matched" if seed % 4 == 1 else bool(seed % 2)), n_min=10)
    scale = 1.0
    if seed % 3 == 0:   # far from the origin
        off = np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), 0.0], np.float32)
        S[:, :, :3] += np.where(S[:, :, 3:4] > 0, off, 0); D[:, :, :3] += np.where(D[:, :, 3:4] > 0, off, 0)
        scale = float(np.abs(off).max()) + 50.0
    s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
    a = rp.default_args(max_points=N, icp_max_iterations=int(rng.choice([20, 50, 100])), icp_stop_mode=("reference", "per_pair")[seed % 2])
    with _lib.options(no_dir_keys=True):
        P0 = utils_hist.estimate_init_pose(a, s, d)
        T0, ev0, i0 = utils_match.hist_icp_eval(a, s, d, return_iterations=True)
    P1 = utils_hist.estimate_init_pose(a, s, d)
    T1, ev1, i1 = utils_match.hist_icp_eval(a, s, d, return_iterations=True)
    p = S[:, :, :3].astype(np.float64)
    mv = lambda M: np.einsum("bij,bnj->bni", M.cpu().numpy().astype(np.float64)[:, :3, :3], p) + M.cpu().numpy().astype(np.float64)[:, None, :3, 3]   # noqa: E731
    dis = float((np.abs(mv(T0) - mv(T1)).max(-1) * (S[:, :, 3] > 0)).max())
    tol = 1e-9 + 4e-7 * scale * (scale > 1.0)
    differ = int((T0 != T1).flatten(1).any(1).sum())
    ok = torch.equal(P0, P1) and int(i0) == int(i1) and dis <= tol and all(torch.allclose(x, y, rtol=1e-5, atol=1e-6) for x, y in zip(ev0, ev1))
    bad += not ok; nbits += differ; npairs += B; worst = max(worst, dis / max(scale, 1.0))
    print(f"seed {seed:3d} B {B:4d} N {N:5d} iters {int(i1):3d} pairs whose transform differs in a bit {differ:3d}, largest displacement {dis:.2e} m (coordinates ~{scale:.0f} m) {'ok' if ok else 'DIFFERENT'}", flush=True)
print(f"different: {bad}; {nbits} of {npairs} transforms differ in some bit; worst displacement relative to the coordinates' magnitude {worst:.2e}")
