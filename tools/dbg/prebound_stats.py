"""Developer tool: what the occupancy pre-bound of the scoring sweeps sees (-DICPFLOW_OCC_STATS build of nn.hip):
  SWEEP_SRC=nn.hip bash tools/dbg/icp_define_build.sh ICPFLOW_OCC_STATS ; ICPFLOW_HIP_LIB=tools/dbg/sweep_1.so python tools/dbg/prebound_stats.py
scans that evaluated it (their block 0), scans it ended, mean ring level of their queries, histogram of (pre-bound / candidate 0's forward mean)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
_lib._L.icpflow_debug_occ_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
def stats():
    buf = np.zeros(12, np.uint64)
    assert _lib._L.icpflow_debug_occ_stats(buf.ctypes.data, 1) == 0
    return [int(v) for v in buf]
for name, B, N, seed, ragged in (("config 2", 256, 1024, 0, False), ("config 4 shard", 1024, 2048, 0, False), ("ragged 600 x 1024", 600, 1024, 31, True), ("ragged 128 x 10000", 128, 10000, 0, True)):
    S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=20) if ragged else synthetic.make_batch(B, N, seed=seed)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    stats()
    utils_match.hist_icp(a, s, d)
    st = stats()
    print(f"{name}: blocks with a pre-bound {st[0]}, ended by it {st[1]}; mean ring level of the queries {st[2] / max(1, st[3]):.2f} (of 3: no target within that many cells); pre-bound / bound in quarters (last: >= 1.75): {st[4:12]}", flush=True)
