"""Developer tool (library built with -DICPFLOW_CERT_STATS, tools/dbg/build_debug.sh): per ICP iteration of BASELINE
config 2 (or B, N from the environment), how many waves ran / searched, how many queries searched, targets scanned."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = int(os.environ.get("B", 256)), int(os.environ.get("N", 1024))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
st = (ctypes.c_ulonglong * 768)()
torch.cuda.synchronize()
_lib._L.icpflow_debug_set_stats_block(int(os.environ.get('PAIR', -1)))
_lib._L.icpflow_debug_cert_stats(st, 1)
T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
torch.cuda.synchronize()
oc = (ctypes.c_ulonglong * 512)()
_lib._L.icpflow_debug_occ_cert(oc)   # (before the reset below)
_lib._L.icpflow_debug_cert_stats(st, 1)
occ = np.array(oc[:], dtype=np.int64).reshape(128, 4)
v = np.array(st[:512], dtype=np.int64).reshape(128, 4)
pr = np.array(st[512:], dtype=np.int64).reshape(128, 2)
print("stop iteration", int(it))
print("  it   waves  scanning  queries scanning  targets/scanning wave   probes  conclusive | without a certificate, of those in an empty cell of the fixed cloud's grid (outside the gate by occupancy), all queries in empty cells")
for k in range(128):
    if v[k, 0] == 0: break
    print(f"  {k:3d} {v[k,0]:6d} {v[k,1]:8d} {v[k,2]:12d}   {v[k,3] / max(v[k,1], 1):8.1f}   {pr[k,0]:8d} {pr[k,1]:8d} | {occ[k,0]:8d} {occ[k,1]:8d} {occ[k,3]:8d}")
