"""Developer tool: the certificates / probes against the plain window scan on many random batches (bit-identical
transforms and iteration counts expected)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match, utils_icp_pytorch3d as icp
from oracle import reference_path as rp
bad = 0
for seed in range(int(os.environ.get("FIRST", 0)), int(os.environ.get("FIRST", 0)) + int(os.environ.get("SEEDS", 40))):
    rng = np.random.default_rng(seed)
    B, N = int(rng.integers(8, 300)), int(rng.choice([96, 300, 700, 1024, 1500, 2048, 3000, 4096]))
    if seed % 5 == 4:   # few large pairs (teams), or many pairs (two workgroups per CU)
        B, N = (int(rng.integers(4, 40)), int(rng.choice([3000, 6000, 10000]))) if seed % 2 else (int(rng.integers(512, 1300)), int(rng.choice([1024, 1500, 2048])))
    S, D, _ = synthetic.make_batch(B, N, seed=1000 * seed, ragged=bool(seed % 2), n_min=10)
    if seed % 3 == 0:   # far from the origin
        off = np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), 0.0], np.float32)
        S[:, :, :3] += np.where(S[:, :, 3:4] > 0, off, 0); D[:, :, :3] += np.where(D[:, :, 3:4] > 0, off, 0)
    s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
    a = rp.default_args(max_points=N, icp_max_iterations=int(rng.choice([20, 50, 100])),
                        icp_stop_mode=("reference", "per_pair")[seed % 2])
    with _lib.options(no_adaptive_windows=True):
        T0, i0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        r0 = icp.iterative_closest_point(s, d, max_iterations=30)
        R0, n0 = r0.RTs.R.clone(), r0.converged.iterations
    T1, i1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    r1 = icp.iterative_closest_point(s, d, max_iterations=30)
    ok = torch.equal(T0, T1) and int(i0) == int(i1) and torch.equal(R0, r1.RTs.R) and n0 == r1.converged.iterations
    bad += not ok
    print(f"seed {seed:3d} B {B:4d} N {N:5d} iters {int(i1):3d}/{n0:3d} {'ok' if ok else 'DIFFERENT'}")
print("different:", bad)
