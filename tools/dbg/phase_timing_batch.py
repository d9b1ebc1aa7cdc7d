"""Developer tool (library built with -DICPFLOW_PHASE_TIMING): shader-clock stamps of the LAST iteration executed by
one workgroup (PAIR, default 197: one of the slow, sliding pairs of BASELINE config 2) inside the real hist_icp flow."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N, PAIR = int(os.environ.get("B", 256)), int(os.environ.get("N", 1024)), int(os.environ.get("PAIR", 197))
S, D, _ = synthetic.make_batch(B, N, seed=0)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
torch.cuda.synchronize()
_lib._L.icpflow_debug_set_stamp_block(PAIR)
order = [(0, "kernel entry"), (1, "queries loaded, scan starts"), (2, "own scan share done"),
         (10, "resolve + x0 reload done"), (3, "moments reduced into LDS"),
         (4, "block barrier passed"), (5, "totals + H formed"), (13, "quartic coefficients"), (14, "newton done"), (6, "kabsch done"), (15, "T, rmse formed"), (9, "history + tally + stop check"), (7, "ring compare, R,T,rmse published"), (8, "loop exit")]
for cap in [int(c) for c in os.environ.get("CAPS", "12,30,46").split(",")]:
    a = rp.default_args(max_points=N, icp_max_iterations=cap)
    T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 16)()
    _lib._L.icpflow_debug_phase_stamps(st)
    v = np.array(st[:16], dtype=np.int64)
    print(f"cap {cap} (stopped after {int(it)}): stamps of the last iteration of pair {PAIR}, total {v[8]-v[0]} shader clocks")
    prev = v[0]
    for idx, name in order:
        if v[idx] == 0: continue
        print(f"   +{v[idx]-prev:8d}  {name}")
        prev = v[idx]
    ws = (ctypes.c_longlong * 256)()
    _lib._L.icpflow_debug_wave_stamps(ws)
    w = np.array(ws[:], dtype=np.int64).reshape(16, 16)
    t0 = w[:, 1].min()
    print("   per wave (relative to the earliest start): start(1), certificates done(11), window known(12), search done(2), resolved(10), moments(3), barrier passed(4), window size")
    for i in range(16):
        if w[i,1] == 0: continue
        print(f"   wave {i:2d}: " + " ".join(f"{w[i,k]-t0:7d}" for k in (1, 11, 12, 2, 10, 3, 4)) + f"   targets {w[i,15]}")
