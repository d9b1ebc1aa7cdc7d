"""Developer tool (-DICPFLOW_SWEEP_CLOCK): per job of the roll-back check sweep -- wall-clock start / end, shader clocks before and in the
window loop, rounds, targets, queries -- on a subset of the ragged real-shape batch (SUBSET, WIDTH as tools/dbg/ragged_subset.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda:0")
B, N = 128, 10000
S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=True, n_min=20)
n = np.minimum((S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1))
idx = np.nonzero(n <= 100)[0] if os.environ.get("SUBSET", "tiny") == "tiny" else np.arange(B)
W = int(os.environ.get("WIDTH", N))
src = torch.from_numpy(np.ascontiguousarray(S[idx][:, :W])).to(dev); dst = torch.from_numpy(np.ascontiguousarray(D[idx][:, :W])).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=W, icp_max_iterations=100, icp_stop_mode="reference")
for _ in range(3): utils_match.hist_icp_eval(a, src, dst)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32768)(); _lib._L.icpflow_debug_sweep_clk(buf)
PER = int(os.environ.get("JOBS_PER_PAIR", 2))      # 2: the check sweep; 12: the scoring sweeps (-DICPFLOW_SWEEP_CLOCK_MODE=0)
v = np.array(buf[:], dtype=np.int64).reshape(4096, 8)[: PER * len(idx)]
v = v[v[:, 1] > 0]
last = v[:, 1].max()
v = v[v[:, 0] > last - 100000]                   # (the last call's records: within a millisecond of the end)
t0 = v[:, 0].min()
print(f"{len(idx)} pairs, width {W}: check sweep spans {(v[:, 1].max() - t0) / 100:.1f} us of wall clock")
print("job: of its block that ended last -- start us, end us | clocks before the loop, in the loop | rounds, targets, queries, block")
order = np.argsort(v[:, 1])
for k in list(order[:6]) + list(order[-10:]):
    r = v[k]
    print(f"  job {k:3d}: {(r[0] - t0) / 100:7.1f} {(r[1] - t0) / 100:7.1f} | {r[2]:8d} {r[3]:8d} | {r[4]:3d} {r[5]:5d} {r[6]:5d} {r[7]:5d}")
