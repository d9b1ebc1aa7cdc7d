"""Developer tool (-DICPFLOW_TAIL_CLOCK): serial / search clocks per iteration of the slowest pairs of the ragged real-shape batch
(SIZES=matched|independent)."""
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
for _ in range(2): T, it = utils_match.hist_icp(a, src, dst, return_iterations=True)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(buf)
v = np.array(buf[:], dtype=np.int64).reshape(1024, 3)[:B]
its = np.maximum(v[:, 2], 1)
print(f"stop after {int(it.item())} iterations; pairs by clocks of member 0 (serial part + search / exchange):")
for k in np.argsort(-(v[:, 0] + v[:, 1]))[:int(os.environ.get('TOP', '30'))]:
    print(f"   pair {k:3d}: {ns[k]:5d} x {nd[k]:5d} points, {v[k, 2]:3d} iterations, per iteration: serial {v[k, 0] / its[k]:7.0f} clocks, search + exchange {v[k, 1] / its[k]:7.0f}")
tot = v[:, 0] + v[:, 1]
print("pairs by executed iterations:", np.bincount(np.minimum(v[:, 2], 100) // 10).tolist(), "(bins of 10)")
print("total clocks per pair: median %.0f, p90 %.0f, max %.0f (%.3f ms at 2.4 GHz)" % (np.median(tot), np.percentile(tot, 90), tot.max(), tot.max() / 2.4e6))
