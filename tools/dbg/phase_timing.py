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
_opts = _lib.options(search=os.environ.get("SEARCH", "auto")); _opts.__enter__()
names = ["entry->scan", "scan (stage+tiles)", "resolve+gate+acc", "block_sum7", "pass2+block_sum9", "kabsch", "pass3+block_sum1", "exit"]
for k in (1, 3, 8, 20, 40):
    icp.iterative_closest_point(src, dst, max_iterations=k)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 16)()
    rc = _lib._L.icpflow_debug_phase_stamps(st)
    v = np.array(st[:16], dtype=np.int64)
    order = [(0, "kernel entry"), (1, "queries loaded, scan starts"), (2, "own scan share done"),
             (10, "resolve + x0 reload done"), (3, "moments reduced into LDS"),
             (4, "block barrier passed"), (5, "totals + H formed"), (13, "quartic coefficients"), (14, "newton done"), (6, "kabsch done"), (15, "T, rmse formed"), (9, "history + tally + stop check"), (7, "ring compare, R,T,rmse published"), (8, "loop exit")]
    print(f"max_iterations={k} (stamps of the LAST iteration, workgroup 0 thread 0), total {v[8]-v[0]} shader clocks")
    prev = v[0]
    for idx, name in order:
        if v[idx] == 0: continue
        print(f"   +{v[idx]-prev:8d}  {name}")
        prev = v[idx]

    if k >= 3:
        ws = (ctypes.c_longlong * 256)()
        _lib._L.icpflow_debug_wave_stamps(ws)
        w = np.array(ws[:], dtype=np.int64).reshape(16, 16)
        t0 = w[:, 1].min()
        print("   per wave (relative to the earliest scan start): start(1), queries loaded(11), window known(12), search done(2), resolved(10), moments(3), barrier passed(4), window size")
        for i in range(16):
            if w[i,1] == 0: continue
            print(f"   wave {i:2d}: " + " ".join(f"{w[i,k]-t0:7d}" for k in (1, 11, 12, 2, 10, 3, 4)) + f"   targets {w[i,15]}")
