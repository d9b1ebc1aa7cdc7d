"""Developer tool: the ragged real-shape batch of bench.py by subsets of its pairs (SUBSET=all|small|big|largest|mid), ten
hist_icp_eval calls -- run under rocprofv3 --kernel-trace and read with tools/dbg/kernel_avg.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
dev = torch.device("cuda:0")
B, N = 128, 10000
S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=("matched" if os.environ.get("SIZES") == "matched" else True), n_min=20)
n = np.minimum((S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1))
sub = os.environ.get("SUBSET", "all")
idx = {"all": np.arange(B), "small": np.nonzero(n <= 1000)[0], "mid": np.nonzero((n > 1000) & (n <= 3000))[0],
       "big": np.nonzero(n > 3000)[0], "largest": np.argsort(-n)[:1], "tiny": np.nonzero(n <= 100)[0],
       "s300": np.nonzero((n > 100) & (n <= 300))[0], "s1000": np.nonzero((n > 300) & (n <= 1000))[0],
       "one_tiny": np.nonzero(n <= 100)[0][:1], "one_s1000": np.nonzero((n > 300) & (n <= 1000))[0][:1]}[sub]
W = int(os.environ.get('WIDTH', N))
src, dst = torch.from_numpy(np.ascontiguousarray(S[idx][:, :W])).to(dev), torch.from_numpy(np.ascontiguousarray(D[idx][:, :W])).to(dev)
N = W
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=100, icp_stop_mode="reference")
utils_match.hist_icp_eval(a, src, dst); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): utils_match.hist_icp_eval(a, src, dst)
torch.cuda.synchronize()
print(f"{sub}: {len(idx)} pairs (smaller cloud: {np.sort(n[idx])[::-1][:8]} ...), hist_icp_eval {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
