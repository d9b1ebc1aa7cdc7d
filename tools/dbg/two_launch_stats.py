"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): the ICP of a batch of a few rounds in ONE launch and in two (icp.hip
icp_split_kernel: the grid of half-CU workgroups drained, the rest on whole CUs) -- when the drain happens, how many pairs the second
launch serves, how long they take there against the same pairs in one launch.  B=1024 N=2048 python tools/dbg/two_launch_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 2048))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
def run(**kw):
    with _lib.options(**kw):
        utils_match.hist_icp(a, s, d)
        prof = _lib.Profile(4)
        with _lib.options(profile=prof, **kw):
            utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize()
        ms, n = prof.collect(); prof.close()
    st = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(st)
    v = np.array(st[:], dtype=np.int64).reshape(1024, 3)[:min(B, 1024)]
    w = (ctypes.c_longlong * 32768)(); _lib._L.icpflow_debug_wg_wall(w)
    w = np.array(w[:], dtype=np.int64).reshape(8192, 4)[:min(B, 1024)]
    t0 = w[:, 0].min()
    return ms / n, v, (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
ms1, v1, s1, e1 = run()
print(f"ONE launch: {ms1:.3f} ms, span by wall clocks {e1.max():.0f} us; unfinished pairs at 50/60/70/80/90 % of it: "
      f"{[int((e1 > f * e1.max()).sum()) for f in (.5, .6, .7, .8, .9)]}; the time at which <= 256 are unfinished: {np.sort(e1)[-257]:.0f} us")
ms2, v2, s2, e2 = run(two_launch=True)
# the pairs of the second launch: the records of their LAST workgroup (start after the first launch's pairs have gone)
order = np.sort(s2)
gap = np.argmax(np.diff(order)[len(order) // 2:]) + len(order) // 2      # the largest jump in the start times of the later half
cut = order[gap + 1]
second = s2 >= cut
print(f"TWO launches: {ms2:.3f} ms, span {e2.max():.0f} us; second launch: {second.sum()} pairs, they start at {s2[second].min():.0f} .. {s2[second].max():.0f} us; "
      f"the first launch's last workgroup leaves at {e2[~second].max():.0f} us")
if second.any():
    dur = e2[second] - s2[second]
    its = v2[second, 2]
    rem1 = e1[second] - cut
    print(f"   second launch: durations mean {dur.mean():.0f} us max {dur.max():.0f}; the same pairs in ONE launch were done {np.mean(e1[second]):.0f} us (mean) / {e1[second].max():.0f} us (last) "
          f"after its start, i.e. {rem1.mean():.0f} / {rem1.max():.0f} us after the time of the cut")
    k = np.argsort(-dur)[:10]
    idx = np.nonzero(second)[0][k]
    for b in idx:
        print(f"   pair {b}: second launch {s2[b]:.0f} -> {e2[b]:.0f} us ({e2[b] - s2[b]:.0f}), total iterations {v2[b, 2]}; in ONE launch {s1[b]:.0f} -> {e1[b]:.0f} us, {v1[b, 2]} iterations, "
              f"{(e1[b] - s1[b]) / max(v1[b, 2], 1):.1f} us per iteration; clocks in the second launch {v2[b, 0] + v2[b, 1]:.3g} (tail {v2[b, 0]:.3g})")
