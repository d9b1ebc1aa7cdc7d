import ctypes, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = 1024, 2048
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
for _ in range(2): T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
torch.cuda.synchronize()
st = (ctypes.c_longlong * 3072)()
_lib._L.icpflow_debug_tail_clock(st)
v = np.array(st[:], dtype=np.int64).reshape(1024, 3)[:B]
tot = v[:, 0] + v[:, 1]
print("stop iteration", int(it), "pairs", B)
print("iterations executed: hist", np.bincount(np.minimum(v[:, 2], 50) // 5).tolist())
print("per-pair clocks: mean %.3g median %.3g max %.3g; sum/512 slots = %.3g clocks (%.2f ms at 2.4 GHz); max pair %.2f ms" % (tot.mean(), np.median(tot), tot.max(), tot.sum() / 512, tot.sum() / 512 / 2.4e6, tot.max() / 2.4e6))
print("tail share %.2f" % (v[:, 0].sum() / tot.sum()))
q = np.percentile(tot, [10, 25, 50, 75, 90, 99]); print("percentiles", (q / 2.4e6).round(3).tolist(), "ms")
# how predictable is a pair's cost before the ICP starts?  its initial residual (mean NN distance under the initial pose)
from icp_flow_amd import utils_hist
T0 = utils_hist.estimate_init_pose(a, s, d)
res = np.zeros(B, np.float32); cnt = np.zeros(B)
for i0 in range(0, B, 64):
    ss, dd = s[i0:i0 + 64], d[i0:i0 + 64]
    moved = ss[:, :, :3] + T0[i0:i0 + 64, None, :3, 3]
    d2 = torch.cdist(moved, dd[:, :, :3])
    d2 = d2.masked_fill(dd[:, None, :, 3] <= 0, 1e9)
    nn = d2.min(dim=2).values
    valid = ss[:, :, 3] > 0
    res[i0:i0 + 64] = ((nn * valid).sum(1) / valid.sum(1)).cpu().numpy()
    cnt[i0:i0 + 64] = valid.sum(1).cpu().numpy()
its = v[:, 2]
print("corr(clocks, initial residual) %.3f, corr(clocks, iterations) %.3f, corr(clocks, points) %.3f" % (np.corrcoef(tot, res)[0, 1], np.corrcoef(tot, its)[0, 1], np.corrcoef(tot, cnt)[0, 1]))
order = np.argsort(-res)
def makespan(order, slots=512):
    import heapq
    h = [0.0] * slots; heapq.heapify(h)
    for b in order: heapq.heappush(h, heapq.heappop(h) + tot[b])
    return max(h) / 2.4e6
print("makespan (ms, 512 slots): as dispatched %.2f, by residual desc %.2f, by true cost desc (LPT) %.2f, by points desc %.2f" % (makespan(np.arange(B)), makespan(order), makespan(np.argsort(-tot)), makespan(np.argsort(-cnt))))
# a second predictor: how ambiguous the six candidate translations are (score of the runner-up over the winner's)
from icp_flow_amd import hist as hip_hist
ex, ey, ez = utils_hist.bin_edges(a)
ratio = np.zeros(B, np.float32); nclose = np.zeros(B)
for i0 in range(0, B, 32):
    ss, dd = s[i0:i0 + 32], d[i0:i0 + 32]
    nb = ss.shape[0]
    h = hip_hist.hist(dd, ss, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
    _, idx = utils_hist.topk_nms(h)
    H, W, Dz = len(ex), len(ey), len(ez)
    exd, eyd, ezd = ex.cuda(), ey.cuda(), ez.cuda()
    t = torch.stack([exd[idx // Dz // W % H], eyd[idx // Dz % W], ezd[idx % Dz]], dim=-1)
    t = torch.cat([t, t.new_zeros(nb, 1, 3)], dim=1)
    sc = torch.empty(nb, 6, device="cuda")
    for k in range(6):
        d2 = torch.cdist(ss[:, :, :3] + t[:, k, None, :], dd[:, :, :3])
        sc[:, k] = torch.minimum(d2.min(dim=2).values.mean(1), d2.min(dim=1).values.mean(1))
    best2 = sc.topk(2, dim=1, largest=False).values
    ratio[i0:i0 + nb] = (best2[:, 1] / best2[:, 0]).cpu().numpy()
    nclose[i0:i0 + nb] = (sc < 1.5 * best2[:, :1]).sum(1).cpu().numpy()
print("corr(clocks, runner-up / winner) %.3f, corr(clocks, candidates within 1.5x) %.3f" % (np.corrcoef(tot, ratio)[0, 1], np.corrcoef(tot, nclose)[0, 1]))
print("makespan by ambiguity (most ambiguous first) %.2f, by candidates within 1.5x %.2f" % (makespan(np.argsort(ratio)), makespan(np.argsort(-nclose))))
