"""Developer tool: does "a scan of the pair survives the occupancy pre-bound" (another candidate with real overlap: an ambiguous,
slide-along pair) predict the pairs whose ICP runs long?  Two builds, two processes, the same batch (config 4's shard):
  MODE=survivors ICPFLOW_HIP_LIB=<-DICPFLOW_OCC_STATS build of nn.hip>   -> gpurun_out/order_survivors.npy
  MODE=clocks    ICPFLOW_HIP_LIB=<-DICPFLOW_TAIL_CLOCK build of icp.hip> -> gpurun_out/order_clocks.npy
  MODE=report    (no GPU work)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mode = os.environ.get("MODE", "report")
out = os.path.join(ROOT, "gpurun_out")
if mode != "report":
    import torch
    from icp_flow_amd import _lib, synthetic, utils_match
    from oracle import reference_path as rp
    B, N = 1024, 2048
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    a = rp.default_args(max_points=N, icp_max_iterations=50)
    s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
    if mode == "survivors":
        buf = np.zeros(1024, np.uint32)
        _lib._L.icpflow_debug_occ_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib._L.icpflow_debug_occ_pairs(buf.ctypes.data, 1)
        utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
        _lib._L.icpflow_debug_occ_pairs(buf.ctypes.data, 1)
        np.save(os.path.join(out, "order_survivors.npy"), buf)
    else:
        utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
        st = (ctypes.c_longlong * 3072)()
        _lib._L.icpflow_debug_tail_clock(st)
        np.save(os.path.join(out, "order_clocks.npy"), np.array(st[:], dtype=np.int64).reshape(1024, 3))
        # candidate predictors, per pair: the mean NN error under the initial pose (what the scoring knows), the clouds' extents
        from icp_flow_amd import utils_hist, utils_helper
        Ti = utils_hist.estimate_init_pose(a, s, d)
        moved = utils_helper.transform_points_batch(s, Ti)
        _, err = utils_helper.nearest_neighbor_batch(moved[:, :, :3], d[:, :, :3])
        e0 = (err * (s[:, :, 3] > 0)).sum(1) / (s[:, :, 3] > 0).sum(1)
        ext = (S[:, :, :3].max(1) - S[:, :, :3].min(1))
        tn = np.linalg.norm(Ti.cpu().numpy()[:, :3, 3], axis=1)
        np.save(os.path.join(out, "order_features.npy"), np.stack([e0.cpu().numpy(), np.sort(ext, 1)[:, 2], np.sort(ext, 1)[:, 1], np.sort(ext, 1)[:, 0], tn], 1))
else:
    sv = np.load(os.path.join(out, "order_survivors.npy")); v = np.load(os.path.join(out, "order_clocks.npy"))
    tot, its = v[:, 0] + v[:, 1], v[:, 2]
    long_ = tot > 2.5 * np.median(tot)
    print(f"pairs {len(tot)}, with a surviving scan {int((sv > 0).sum())}, long pairs (> 2.5 x the median clocks) {int(long_.sum())}; long AND flagged {int((long_ & (sv > 0)).sum())}")
    print(f"mean clocks: flagged {tot[sv > 0].mean():.3g}, others {tot[sv == 0].mean():.3g}; iterations: flagged {its[sv > 0].mean():.1f}, others {its[sv == 0].mean():.1f}")
    if os.path.exists(os.path.join(out, "order_features.npy")):
        F = np.load(os.path.join(out, "order_features.npy"))
        from scipy.stats import spearmanr
        for k, name in enumerate(("mean NN error under the initial pose", "longest extent", "middle extent", "shortest extent", "|initial translation|")):
            print(f"  rank correlation of the pair's clocks with {name}: {spearmanr(F[:, k], tot).correlation:+.3f}; with its iterations: {spearmanr(F[:, k], its).correlation:+.3f}")
    order = np.argsort(-(sv > 0).astype(int), kind="stable")
    # makespan model: 512 slots, pairs in the given order, a free slot takes the next pair
    def makespan(seq):
        import heapq
        h = [0.0] * 512; heapq.heapify(h)
        for k in seq: heapq.heappush(h, heapq.heappop(h) + tot[k] / 2.4e6)
        return max(h)
    print(f"makespan model (ms, no helpers): as it comes {makespan(range(len(tot))):.3f}, flagged first {makespan(order):.3f}, longest first {makespan(np.argsort(-tot)):.3f}")
