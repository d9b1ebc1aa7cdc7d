"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): the timeline of the persistent ICP launch of config 4's shard -- resident owners over
the span, when the pairs start, microseconds per iteration and helper passes of the pairs that end the launch."""
import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = 1024, 2048
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
with _lib.options():
    utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
    ph = (ctypes.c_ulonglong * 1024)(); _lib._L.icpflow_debug_pair_help(ph, 1)
    hc = (ctypes.c_ulonglong * 4096)(); _lib._L.icpflow_debug_pair_hclk(hc, 1)
    utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
_lib._L.icpflow_debug_pair_help(ph, 0); _lib._L.icpflow_debug_pair_hclk(hc, 0)
ph = np.array(ph[:], dtype=np.int64); hc = np.array(hc[:], dtype=np.int64).reshape(1024, 4)
st = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(st)
v = np.array(st[:], dtype=np.int64).reshape(1024, 3)
w = (ctypes.c_longlong * 32768)(); _lib._L.icpflow_debug_wg_wall(w)
w = np.array(w[:], dtype=np.int64).reshape(8192, 4)[:B]
t0 = w[:, 0].min(); start = (w[:, 0] - t0) / 100.; end = (w[:, 1] - t0) / 100.
its = v[:, 2]
order = np.argsort(-end)[:24]
print("span", end.max())
grid = np.linspace(0, end.max(), 21)
print("resident owners at 0,5,..100%:", [(int((start <= t).sum() - (end <= t).sum())) for t in grid])
print("last pair to START at", start.max(), "us; pairs started after 300us:", (start > 300).sum(), "after 500us:", (start > 500).sum())
for b in order:
    print(f"pair {b}: start {start[b]:.0f} end {end[b]:.0f} its {its[b]} us/it {(end[b]-start[b])/its[b]:.1f} helper passes received {ph[b]} (of {its[b]*3} possible) owner waited {hc[b,3]/100:.0f} us")
long = its >= 40
print("pairs with >= 40 iterations:", long.sum(), "start times quantiles", np.quantile(start[long], [0, .25, .5, .75, 1]).round(0).tolist(), "us/it mean", ((end-start)/its)[long].mean().round(1),
      "helper passes share", (ph[long].sum() / (its[long].sum()*3)).round(3))
