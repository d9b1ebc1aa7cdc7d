"""Developer tool: match_eval time per batch shape with the library selected by ICPFLOW_HIP_LIB (sweep threshold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
out = []
for B, N, rag in ((256, 1024, False), (256, 512, False), (1024, 1500, False), (1024, 2048, False), (600, 2048, True), (128, 1024, True), (81, 2048, True)):
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=rag, n_min=20)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    T = utils_match.hist_icp(a, s, d)
    best = 1e9
    for rep in range(3):
        for _ in range(3): utils_match.match_eval(a, s, d, T)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): utils_match.match_eval(a, s, d, T)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20 * 1e3)
    out.append(f"{B}x{N}{'r' if rag else ''}: {best:.4f}")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), "match_eval ms:", " | ".join(out))
