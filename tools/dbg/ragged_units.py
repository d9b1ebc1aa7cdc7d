"""Developer tool (-DICPFLOW_TAIL_CLOCK): per iteration of chosen pairs of the ragged real-shape batch (SIZES=matched|independent,
PAIRS=9,81,126): the search's critical path (sum over passes of the slowest wave's unit, shader clocks, thousands), the
largest window and the lanes that searched.  The record is written by every member of the pair's team (last writer wins)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
B, N = 128, 10000
sizes = "matched" if os.environ.get("SIZES", "matched") == "matched" else True
S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=sizes, n_min=20)
ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=100, icp_stop_mode="reference")
buf = (ctypes.c_longlong * 8192)()
np.set_printoptions(linewidth=250)
for pair in [int(x) for x in os.environ.get("PAIRS", "9,81,126").split(",")]:
    _lib._L.icpflow_debug_unit_clk(buf, pair)
    utils_match.hist_icp(a, src, dst)
    torch.cuda.synchronize()
    _lib._L.icpflow_debug_unit_clk(buf, -1)
    u = np.array(buf[:], dtype=np.int64).reshape(64, 8, 16)
    wb = (ctypes.c_int * 16384)(); _lib._L.icpflow_debug_unit_win(wb)
    win = np.array(wb[:], dtype=np.int64).reshape(64, 8, 16, 2)
    print(f"pair {pair}: {ns[pair]} x {nd[pair]} points")
    crit = u.max(2).sum(1) / 1e3
    print("  search critical path per iteration (k clocks):", np.round(crit[:56], 0).astype(int).tolist())
    print("  passes with work per iteration:", (u.max(2) > 0).sum(1)[:56].tolist())
    print("  largest window per iteration:", win[:, :, :, 0].max((1, 2))[:56].tolist())
    print("  lanes searching per iteration (all waves, passes):", win[:, :, :, 1].sum((1, 2))[:56].tolist())
    for it in [int(x) for x in os.environ.get("SHOW", "").split(",") if x]:
        print(f"  iteration {it}: clocks (k) per (pass, wave):\n{np.round(u[it, :4, :12] / 1e3, 0).astype(int)}\n  window per (pass, wave):\n{win[it, :4, :12, 0]}\n  lanes searching:\n{win[it, :4, :12, 1]}")
