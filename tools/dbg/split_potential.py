"""Developer tool: what pipelining sub-batches could buy -- config 4's shard (1024 x 2048) and all 8192 pairs as ONE hist_icp call
against hist_icp_many over 2 / 4 / 8 sub-batches (each with its own batch rule: the potential only, not the same registration)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
dev = torch.device("cuda:0")
for B, N, reps in ((1024, 2048, 8), (8192, 2048, 3), (256, 1024, 20)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    def t(fn):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    out = [f"one call {t(lambda: utils_match.hist_icp(a, s, d)):.3f} ms"]
    for k in (2, 4, 8):
        if B // k < 64: continue
        ss = [x.contiguous() for x in s.chunk(k)]; dd = [x.contiguous() for x in d.chunk(k)]
        out.append(f"{k} sub-batches in one call {t(lambda: utils_match.hist_icp_many(a, ss, dd)):.3f} ms")
    print(f"{B} x {N}: " + " | ".join(out))
