"""Developer tool: ICP launch / step time of a few large shapes with the library selected by ICPFLOW_HIP_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
out = []
for B, N, reps in ((256, 1024, 30), (1024, 2048, 8), (8192, 2048, 3), (2048, 1024, 8), (600, 2048, 8)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    prof = _lib.Profile(64)
    utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
    with _lib.options(profile=prof):
        t = time.perf_counter()
        for _ in range(reps): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
    icp, n = prof.collect()
    out.append(f"{B}x{N}: step {ms:.3f} icp {icp / n:.3f}")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), " | ".join(out))
