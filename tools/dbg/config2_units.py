"""Developer tool (-DICPFLOW_TAIL_CLOCK): config 2 (256 x 2048), chosen pairs (PAIRS=93,215,47): per iteration the clocks of every
wave's search unit per pass -- the slowest wave against the mean wave (what balancing the waves of a workgroup could give)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
B, N = 256, 2048
S, D, _ = synthetic.make_batch(B, N, seed=0)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=100, icp_stop_mode="reference")
buf = (ctypes.c_longlong * 8192)()
np.set_printoptions(linewidth=250)
for pair in [int(x) for x in os.environ.get("PAIRS", "93,215,47").split(",")]:
    _lib._L.icpflow_debug_unit_clk(buf, pair)
    utils_match.hist_icp(a, src, dst)
    torch.cuda.synchronize()
    _lib._L.icpflow_debug_unit_clk(buf, -1)
    u = np.array(buf[:], dtype=np.int64).reshape(64, 8, 16)
    wb = (ctypes.c_int * 16384)(); _lib._L.icpflow_debug_unit_win(wb)
    win = np.array(wb[:], dtype=np.int64).reshape(64, 8, 16, 2)
    its = int((u.max((1, 2)) > 0).sum())
    per_wave = u.sum(1)                     # [it, wave]: both passes of a wave
    print(f"pair {pair}: {its} iterations; per iteration (k clocks): slowest wave {per_wave[:its].max(1).mean() / 1e3:.1f}, mean wave "
          f"{per_wave[:its].mean() / 1e3:.1f}, fastest {per_wave[:its].min(1).mean() / 1e3:.1f}; "
          f"sum over iterations: slowest {per_wave[:its].max(1).sum() / 1e3:.0f} k, mean {per_wave[:its].mean(1).sum() / 1e3:.0f} k")
    print("  slowest wave per iteration (k):", np.round(per_wave[:its].max(1) / 1e3, 1).tolist())
    print("  mean wave per iteration (k):   ", np.round(per_wave[:its].mean(1) / 1e3, 1).tolist())
    for it in [int(x) for x in os.environ.get("SHOW", "2,10,25").split(",") if x]:
        print(f"  iteration {it}: clocks (k) per (pass, wave):\n{np.round(u[it, :2] / 1e3, 1)}\n  window per (pass, wave):\n{win[it, :2, :, 0]}\n  lanes searching:\n{win[it, :2, :, 1]}")
