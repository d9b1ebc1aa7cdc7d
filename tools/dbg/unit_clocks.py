"""Developer tool (-DICPFLOW_TAIL_CLOCK): shader clocks per (iteration, pass, wave) of one pair of config 4's shard, helpers off."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 2048))
S, D, _ = synthetic.make_batch(B, N, seed=0)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
s, d = torch.from_numpy(S).cuda(), torch.from_numpy(D).cuda()
buf = (ctypes.c_longlong * 8192)()
np.set_printoptions(linewidth=220)
for pair in [int(x) for x in os.environ.get("PAIRS", "751,915").split(",")]:
    _lib._L.icpflow_debug_unit_clk(buf, pair)
    with _lib.options(no_helpers=True):
        utils_match.hist_icp(a, s, d)
    torch.cuda.synchronize()
    _lib._L.icpflow_debug_unit_clk(buf, -1)
    u = np.array(buf[:], dtype=np.int64).reshape(64, 8, 16)[:, :int(os.environ.get("PASSES", 4)), :int(os.environ.get("WAVES", 8))]
    wb = (ctypes.c_int * 16384)(); _lib._L.icpflow_debug_unit_win(wb)
    win = np.array(wb[:], dtype=np.int64).reshape(64, 8, 16, 2)[:, :int(os.environ.get("PASSES", 4)), :int(os.environ.get("WAVES", 8))]
    print(f"pair {pair}: clocks per (pass, wave) in thousands")
    for it in [int(x) for x in os.environ.get("ITS", "0,2,5,10,15,20,25,30,35,40,45").split(",")]:
        m = u[it]
        if m.sum() == 0: continue
        if it in [int(x) for x in os.environ.get("SHOW", "0,10,20,30,40").split(",")]:
            print(f"   iteration {it}, targets in the window per (pass, wave):\n{win[it, :, :, 0]}\n   lanes that searched:\n{win[it, :, :, 1]}\n   clocks (k):\n{np.round(m / 1e3, 0).astype(int)}")
        print(f" iteration {it}: sum over passes per wave {np.round(m.sum(0) / 1e3, 1)}; max unit {m.max() / 1e3:.1f}; slowest wave {m.sum(0).max() / 1e3:.1f}; per pass max over waves {np.round(m.max(1) / 1e3, 1)}")
