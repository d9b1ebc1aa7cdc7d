"""Developer tool: how often the roll-back check takes its initial-pose sum from the scoring, and how often it has to scan for
itself because the picked candidate's forward scan was pruned -- counters of a -DICPFLOW_REUSE_STATS build (pose.hip select_kernel):
  SWEEP_SRC=pose.hip bash tools/dbg/icp_define_build.sh ICPFLOW_REUSE_STATS   (build container)
  ICPFLOW_HIP_LIB=tools/dbg/sweep_1.so python tools/dbg/check_reuse_stats.py  (GPU box)
Every shape also runs with ICPFLOW_OPT_NO_CHECK_REUSE and with every scoring scan run to its end: same poses bit for bit."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
_lib._L.icpflow_debug_reuse_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
def stats(reset=1):
    buf = np.zeros(4, np.uint64)
    assert _lib._L.icpflow_debug_reuse_stats(buf.ctypes.data, reset) == 0
    return [int(v) for v in buf]
shapes = [("config 2", 256, 1024, 0, False), ("config 4 shard", 1024, 2048, 0, False), ("ragged 600 x 1024", 600, 1024, 31, True),
          ("ragged 128 x 10000", 128, 10000, 0, True), ("ragged 90 x 2048", 90, 2048, 5, True)]
shapes += [(f"ragged 300 x 1500 seed {k}", 300, 1500, 100 + k, True) for k in range(6)]
shapes += [("8 x 1100 with two backward winners (tests/test_gpu_fullsize.py)", 8, 1100, 41, True)]
tot = [0, 0, 0, 0]
for name, B, N, seed, ragged in shapes:
    S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=300 if "winners" in name else 20) if ragged else synthetic.make_batch(B, N, seed=seed)
    if "winners" in name:   # pairs whose pick wins through its backward mean while its forward scan is pruned
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_fullsize import _backward_winner_pair
        for b, sd in ((2, 0), (5, 1)): S[b], D[b] = _backward_winner_pair(N, sd)
    if "seed" in name:   # unequal partners: a candidate may win through its backward mean
        rng = np.random.default_rng(seed)
        for b in range(0, B, 3):
            n = int((D[b, :, 3] > 0).sum()); keep = max(20, int(n * rng.uniform(0.2, 0.6)))
            D[b, keep:, 0:3] = 1e8; D[b, keep:, 3] = 0.0
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    stats()
    T1 = utils_match.hist_icp(a, s, d)
    st = stats()
    with _lib.options(no_check_reuse=True):
        T0 = utils_match.hist_icp(a, s, d)
    with _lib.options(no_score_prune=True):
        T2 = utils_match.hist_icp(a, s, d)
    st2 = stats()
    tot = [x + y for x, y in zip(tot, st)]
    print(f"{name}: from the scoring {st[0]}, scanned by the check (pruned scan) {st[1]}, roll-backs {st[3]}; identical to NO_CHECK_REUSE {torch.equal(T0, T1)}, "
          f"to unpruned scoring {torch.equal(T2, T1)} (unpruned: from the scoring {st2[0] - 0}, scanned {st2[1]}, no offer {st2[2]})", flush=True)
print(f"all shapes: from the scoring {tot[0]}, scanned by the check {tot[1]} ({100.0 * tot[1] / max(1, tot[0] + tot[1]):.2f} %), roll-backs {tot[3]}")
