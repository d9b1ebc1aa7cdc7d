"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): per-pair shader clocks of the first 1024 pairs of a B x N batch
(B up to 8192), next to the wall time of the ICP launch: where a batch much larger than the GPU loses its time."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
N = int(os.environ.get("N", 2048))
for B in [int(x) for x in os.environ.get("BS", "1024,2048,8192").split(",")]:
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    a = rp.default_args(max_points=N, icp_max_iterations=50)
    s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
    prof = _lib.Profile(8)
    utils_match.hist_icp(a, s, d)
    hs = (ctypes.c_ulonglong * 8)()
    _lib._L.icpflow_debug_help_stats(hs, 1)
    with _lib.options(profile=prof):
        T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
    torch.cuda.synchronize()
    ms, n = prof.collect()
    _lib._L.icpflow_debug_help_stats(hs, 0)
    print(f"helpers' side: {hs[5]} passes, {hs[4] / max(hs[5], 1):.0f} shader clocks per pass (state read .. sums out), {hs[6] / 100.0 / max(hs[5], 1):.2f} us waiting for the next state per pass")
    print(f"helpers: joined {hs[0]}, found the pair finished {hs[3]}, passes taken from helpers {hs[1]}, owner waves waited {hs[2] / 100.0 / max(hs[1], 1):.2f} us per pass on average")
    st = (ctypes.c_longlong * 3072)()
    _lib._L.icpflow_debug_tail_clock(st)
    v = np.array(st[:], dtype=np.int64).reshape(1024, 3)
    tot = v[:, 0] + v[:, 1]
    print(f"B {B}: icp launch {ms / n:.3f} ms, stop iteration {int(it)}; first 1024 pairs: clocks mean {tot.mean():.3g} median {np.median(tot):.3g} "
          f"max {tot.max():.3g}; sum/512 slots {tot.sum() / 512 / 2.4e6:.3f} ms -> x{B // 1024} = {tot.sum() / 512 / 2.4e6 * B / 1024:.3f} ms; "
          f"iterations mean {v[:, 2].mean():.1f}; clocks per iteration {tot.sum() / max(v[:, 2].sum(), 1):.0f}")
    w = (ctypes.c_longlong * 32768)()
    _lib._L.icpflow_debug_wg_wall(w)
    w = np.array(w[:], dtype=np.int64).reshape(8192, 4)[:B]
    t0 = w[:, 0].min()
    start, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0      # microseconds
    dur = end - start
    clk = tot[:min(B, 1024)] / np.maximum(dur[:min(B, 1024)], 1e-9) / 1e3   # GHz (loop clocks / wall)
    print(f"   wall: launch span {end.max() / 1e3:.3f} ms; workgroup durations mean {dur.mean():.1f} us max {dur.max():.1f}; sum of durations / span = "
          f"{dur.sum() / end.max():.1f} workgroups resident on average; shader clock (loop clocks / wall) median {np.median(clk):.2f} GHz "
          f"min {clk.min():.2f} max {clk.max():.2f}")
    ev = np.sort(np.concatenate([start, end]))
    grid = np.linspace(0, end.max(), 11)
    res = [(int((start <= t).sum() - (end <= t).sum())) for t in grid]
    print("   resident workgroups at 0, 10, ... 100 % of the span:", res)
    cu = (w[:, 2] >> 8) & 0xf; se = (w[:, 2] >> 13) & 0x7; sh = (w[:, 2] >> 12) & 1; xcc = w[:, 3] & 0xf
    first = np.argsort(start)[:512]
    key = xcc[first] * 1000 + se[first] * 100 + sh[first] * 16 + cu[first]
    u, c = np.unique(key, return_counts=True)
    print(f"   the first 512 workgroups to start sit on {len(u)} distinct (xcc, se, sh, cu); per-CU counts histogram {np.bincount(c).tolist()}; started within {start[first].max():.1f} us")
