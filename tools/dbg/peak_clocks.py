"""Developer tool: phases of hist_peaks_kernel per workgroup (build with -DICPFLOW_PEAK_CLOCK), config 2's batch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50, icp_stop_mode="reference")
for _ in range(3): utils_match.hist_icp(a, s, d)
torch.cuda.synchronize()
buf = np.zeros((1024, 8), np.uint64)
_lib._L.icpflow_debug_peak_clock.argtypes = [ctypes.c_void_p]
assert _lib._L.icpflow_debug_peak_clock(buf.ctypes.data) == 0
c = buf[:256].astype(np.int64)
t0 = c[:, 0].min()
print("blocks start %.1f .. %.1f us; end %.1f .. %.1f us after the first" % ((c[:, 0].min() - t0) / 100, (c[:, 0].max() - t0) / 100, (c[:, 6].min() - t0) / 100, (c[:, 6].max() - t0) / 100))
names = ["fill", "pass z", "pass y", "pass x", "waves' top k", "wave 0's pick"]
for k, n in enumerate(names):
    dt = (c[:, k + 1] - c[:, k]) / 100
    print(f"  {n:14s} median {np.median(dt):6.2f} us   max {dt.max():6.2f}")
