"""Developer tool: per-phase shader-clock breakdown of one ICP iteration (workgroup 0).
Build + run on the GPU box:
  hipcc ... -DICPFLOW_PHASE_TIMING -o tools/dbg/libicpflow_phase.so ; ICPFLOW_HIP_LIB=... python tools/dbg/phase_timing.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_icp_pytorch3d as icp
B, N = int(os.environ.get("B", 256)), int(os.environ.get("N", 1024))
S, D, Tt = synthetic.make_batch(B, N, seed=0)
src = torch.from_numpy(S); dst = torch.from_numpy(D)
for i in range(B):   # pre-align so that there are inliers (like after the histogram init)
    Ti = torch.from_numpy(Tt[i]); src[i, :, :3] = src[i, :, :3] @ Ti[:3, :3].T + Ti[:3, 3] + torch.tensor([0.03, -0.02, 0.01])
src, dst = src.cuda(), dst.cuda()
names = ["entry->scan", "scan (stage+tiles)", "resolve+gate+acc", "block_sum7", "pass2+block_sum9", "kabsch", "pass3+block_sum1", "exit"]
for k in (1, 2, 3):
    icp.iterative_closest_point(src, dst, max_iterations=k)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 16)()
    rc = _lib._L.icpflow_debug_phase_stamps(st)
    v = np.array(st[:9], dtype=np.int64)
    d = np.diff(v)
    print(f"max_iterations={k} (stamps of the LAST iteration), total {v[8]-v[0]} clk")
    for n, x in zip(names, d): print(f"   {n:22s} {x:8d} clk  {x/100.0:8.2f} us@100MHz")
