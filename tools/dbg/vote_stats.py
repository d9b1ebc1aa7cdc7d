"""Developer tool (library built with -DICPFLOW_VOTE_STATS): what the sorted vote (hist.hip hist_vote_sorted_kernel) evaluates --
per shape, the (row, target) evaluations its z / u windows let through against the brute-force n_x * n_y, how many of them vote
(fall inside the box), and the vote's time without the counters (the product library, ICPFLOW_HIP_LIB_PLAIN).  VERDICT r5 item 6."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_hist
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
for name, B, N, ragged in (("config 2 (256 x 1024)", 256, 1024, False), ("config 4's shard (1024 x 2048)", 1024, 2048, False),
                           ("ragged matched 128 x 10000", 128, 10000, "matched"), ("ragged independent 128 x 10000", 128, 10000, True)):
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=20) if ragged else synthetic.make_batch(B, N, seed=0)
    a = rp.default_args(max_points=N)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    ns, nd = (S[:, :, 3] > 0).sum(1).astype(np.float64), (D[:, :, 3] > 0).sum(1).astype(np.float64)
    brute = float((ns * nd).sum())
    z = (ctypes.c_ulonglong * 4)()
    utils_hist.estimate_init_pose(a, s, d); torch.cuda.synchronize()
    _lib._L.icpflow_debug_vote_stats(z, 1)
    utils_hist.estimate_init_pose(a, s, d); torch.cuda.synchronize()
    _lib._L.icpflow_debug_vote_stats(z, 0)
    ev, votes, steps, waves = [float(v) for v in z]
    print(f"{name}: brute force {brute:.3g} (row, target) pairs; the windows let {ev:.3g} through ({ev / brute:.1%}); {votes:.3g} vote ({votes / brute:.1%} of brute force, "
          f"{votes / max(ev, 1):.1%} of what the windows let through); wave steps (targets visited by a wave) {steps:.3g}, i.e. {ev / max(steps, 1):.1f} valid rows per step; "
          f"{waves:.3g} (wave, window) visits, {steps / max(waves, 1):.0f} targets each", flush=True)
