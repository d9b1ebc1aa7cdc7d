"""Developer tool (library built with -DICPFLOW_TAIL_CLOCK): per pair of BASELINE config 2, the shader clocks wave 0 spent
in the serial tail (block barrier -> (R, T) published) and in the rest of the iteration loop, two clock reads per iteration.
SPLIT=1 with a library built with -DICPFLOW_TAIL_CLOCK -DICPFLOW_TAIL_SPLIT: the tail of one sliding pair phase by phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_match
from oracle import reference_path as rp
B, N = int(os.environ.get("B", 256)), int(os.environ.get("N", 1024))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = rp.default_args(max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
for _ in range(2):
    T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
torch.cuda.synchronize()
st = (ctypes.c_longlong * 3072)()
_lib._L.icpflow_debug_tail_clock(st)
v = np.array(st[:], dtype=np.int64).reshape(1024, 3)[:B]
order = np.argsort(-(v[:, 0] + v[:, 1]))
print("stop iteration", int(it))
print("slowest pairs: pair, iterations executed, tail clocks (per iteration), rest of the loop (per iteration), total")
for b in order[:12]:
    n = max(v[b, 2], 1)
    print(f"  {b:4d} {v[b,2]:4d}   {v[b,0]:9d} ({v[b,0]/n:7.0f})   {v[b,1]:9d} ({v[b,1]/n:7.0f})   {v[b,0]+v[b,1]:9d}")
print("all pairs: tail %.3g clocks, rest %.3g clocks" % (v[:, 0].sum(), v[:, 1].sum()))
if not os.environ.get("SPLIT"):
    sys.exit(0)
sp = (ctypes.c_longlong * 16384)()
_lib._L.icpflow_debug_tail_split(sp)
w = np.array(sp[:], dtype=np.int64).reshape(1024, 16)[:B]
names = {1: "top of loop -> queries loaded", 11: "certificates + probes", 12: "window", 2: "scan", 10: "resolve + records", 3: "moments", 4: "block barrier", 5: "totals + H", 13: "quartic coefficients",
         14: "newton", 6: "adjugate + rotation", 15: "T, rmse", 9: "history + tally + stop check", 7: "cycle detection + publish", 8: "loop exit"}
b = int(order[2])
print(f"pair {b}: clocks per iteration between the stamps of thread 0 (each stamp costs a clock read + an LDS update)")
for k in (1, 11, 12, 2, 10, 3, 4, 5, 13, 14, 6, 15, 9, 7):
    print(f"   {names[k]:34s} {w[b, k] / max(v[b, 2], 1):8.0f}")
