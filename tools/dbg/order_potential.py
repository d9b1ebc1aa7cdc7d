"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): what the ORDER in which a batch larger than the GPU deals its pairs to the
persistent ICP grid is worth -- config 4's shard as it comes, with its pairs permuted longest first (by the shader clocks of a
first run: the best any predictor could do) and shortest first."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 2048))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
def run(S, D, reps=5):
    s, d = torch.from_numpy(np.ascontiguousarray(S)).cuda(), torch.from_numpy(np.ascontiguousarray(D)).cuda()
    utils_match.hist_icp(a, s, d)
    prof = _lib.Profile(reps + 2)
    import time
    torch.cuda.synchronize(); t = time.perf_counter()
    with _lib.options(profile=prof):
        for _ in range(reps): T = utils_match.hist_icp(a, s, d)
    torch.cuda.synchronize(); step = (time.perf_counter() - t) / reps * 1e3
    ms, n = prof.collect(); prof.close()
    st = (ctypes.c_longlong * 3072)()
    _lib._L.icpflow_debug_tail_clock(st)
    v = np.array(st[:], dtype=np.int64).reshape(1024, 3)
    return step, ms / n, v[:min(B, 1024), 0] + v[:min(B, 1024), 1], v[:min(B, 1024), 2], T
step, icp, tot, its, T0 = run(S, D)
print(f"as it comes: step {step:.3f} ms, ICP launch {icp:.3f} ms; clocks per pair mean {tot.mean():.3g} max {tot.max():.3g}; iterations mean {its.mean():.1f}, >= 40: {(its >= 40).sum()} pairs; sum / 512 slots = {tot.sum() / 512 / 2.4e6:.3f} ms, longest pair {tot.max() / 2.4e6:.3f} ms")
orders = [("longest first", np.argsort(-tot, kind="stable")), ("shortest first", np.argsort(tot, kind="stable"))]
feat = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "order_features.npy")
if os.path.exists(feat) and B == 1024:   # (tools/dbg/order_predictor.py MODE=clocks: what a launch could know beforehand)
    F = np.load(feat)
    orders += [("largest initial NN error first", np.argsort(-F[:, 0], kind="stable")), ("smallest extent first", np.argsort(F[:, 1], kind="stable")),
               ("a random order", np.random.default_rng(0).permutation(1024))]
for name, perm in orders:
    full = np.concatenate([perm, np.arange(len(perm), B)])
    step, icp, _, _, T = run(S[full], D[full])
    same = torch.equal(T.cpu(), T0.cpu()[torch.from_numpy(full)])
    print(f"{name}: step {step:.3f} ms, ICP launch {icp:.3f} ms; same transforms (permuted) {same}")
