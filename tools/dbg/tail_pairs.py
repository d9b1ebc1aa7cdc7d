"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): the last pairs of config 4's shard -- start, end, iterations,
time per iteration -- with and without helpers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
B, N = int(os.environ.get("B", 1024)), 2048
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
for flags in ({}, {"no_helpers": True}):
    with _lib.options(**flags):
        utils_match.hist_icp(a, s, d)
        prof = _lib.Profile(8)
        ph = (ctypes.c_ulonglong * 1024)(); _lib._L.icpflow_debug_pair_help(ph, 1)
        hc = (ctypes.c_ulonglong * 4096)(); _lib._L.icpflow_debug_pair_hclk(hc, 1)
        with _lib.options(profile=prof):
            utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize()
        ms, n = prof.collect()
    st = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(st)
    v = np.array(st[:], dtype=np.int64).reshape(1024, 3)
    w = (ctypes.c_longlong * 32768)(); _lib._L.icpflow_debug_wg_wall(w)
    w = np.array(w[:], dtype=np.int64).reshape(8192, 4)[:min(B, 1024)]
    t0 = w[:, 0].min()
    start, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    its = v[:len(w), 2]
    _lib._L.icpflow_debug_pair_help(ph, 0); helped = np.array(ph[:], dtype=np.int64)[:len(w)]
    _lib._L.icpflow_debug_pair_hclk(hc, 0); hck = np.array(hc[:], dtype=np.int64).reshape(1024, 4)
    print(f"flags {flags}: icp launch {ms / n:.3f} ms; last ticket taken at {start.max():.0f} us; span {end.max():.0f} us")
    order = np.argsort(-end)[:16]
    for k in order:
        print(f"   pair {k:4d}: start {start[k]:7.0f} end {end[k]:7.0f} us, {its[k]:3d} iterations, {(end[k] - start[k]) / max(its[k], 1):6.1f} us / iteration, passes from helpers {helped[k]} (= {helped[k] / 3.0:.0f} fully helped iterations); clocks / iteration: serial part {v[k, 0] / max(its[k], 1):.0f}, search {v[k, 1] / max(its[k], 1):.0f}; helpers: {hck[k, 0] / max(hck[k, 1], 1):.0f} clocks per pass, waited {hck[k, 2] / 100.0 / max(hck[k, 1], 1):.1f} us per pass for the next state; owner waited {hck[k, 3] / 100.0 / max(helped[k], 1):.1f} us per helped pass")
    per = (end - start) / np.maximum(its, 1)
    early = start < 0.3 * start.max()
    slow = per > 1.5 * np.median(per)
    print(f"   pairs slower than 1.5 x the median per iteration: {int(slow.sum())} of {len(per)}; they hold {100 * ((end - start)[slow].sum() / (end - start).sum()):.0f} % of the workgroup time; "
          f"of the 32 pairs that end last, {int(slow[np.argsort(-end)[:32]].sum())} are such pairs")
    print(f"   us / iteration: pairs started in the first 30 % of the tickets: median {np.median(per[early]):.1f}; all: median {np.median(per):.1f}; "
          f"pairs ending in the last 30 % of the span: median {np.median(per[end > 0.7 * end.max()]):.1f}")
