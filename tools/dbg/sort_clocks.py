"""Developer tool: per-block phases of chunk_sort_kernel (build with -DICPFLOW_SORT_CLOCK) on the ragged batch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
S, D, _ = synthetic.make_batch(128, 10000, seed=0, ragged=True, n_min=20)
s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=10000, icp_max_iterations=100, icp_stop_mode="reference")
for _ in range(3): utils_match.hist_icp(a, s, d)
torch.cuda.synchronize()
lib = _lib._L
buf = np.zeros((2, 4096, 6), np.uint64)
lib.icpflow_debug_sort_clock.argtypes = [ctypes.c_void_p]
assert lib.icpflow_debug_sort_clock(buf.ctypes.data) == 0
for mode in (0, 1):
    c = buf[mode].astype(np.int64)
    live = c[:, 4] > 0
    t0 = c[c[:, 0] > 0, 0].min()
    print(f"mode {mode}: blocks stamped {int((c[:, 0] > 0).sum())}, sorting {int(live.sum())}; kernel span {(c[live, 4].max() - t0) / 100:.1f} us")
    rows = c[live]
    order = np.argsort(-(rows[:, 4] - rows[:, 0]))
    print("   n     start    params   load    sort    write   (us; the 12 longest blocks, then the median)")
    for r in list(rows[order[:12]]) + [rows[order[len(order) // 2]]]:
        print(f"  {r[5]:6d} {(r[0] - t0) / 100:8.1f} {(r[1] - r[0]) / 100:8.1f} {(r[2] - r[1]) / 100:7.1f} {(r[3] - r[2]) / 100:7.1f} {(r[4] - r[3]) / 100:7.1f}")
    print("   last block to START: %.1f us" % ((c[c[:, 0] > 0, 0].max() - t0) / 100))
