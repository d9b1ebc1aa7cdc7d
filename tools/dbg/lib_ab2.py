import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
out = []
for sh in os.environ.get("SHAPES", "1024x2048,1500x1500,2048x1100,r900x2048,600x2048").split(","):
    ragged = sh.startswith("r"); B, N = map(int, sh.lstrip("r").split("x"))
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=30) if ragged else synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    T, it = utils_match.hist_icp(a, s, d, return_iterations=True); torch.cuda.synchronize()
    prof = _lib.Profile(16)
    with _lib.options(profile=prof):
        t = time.perf_counter()
        for _ in range(8): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 8 * 1e3
    icp, n = prof.collect(); prof.close()
    out.append(f"{sh}: step {ms:.3f} icp {icp / n:.3f} it {int(it)} sum {float(T.double().sum()):.10f}")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), " | ".join(out), flush=True)
