"""Developer tool: what the HIP events around the ICP launch (icpflow_profile_t: bench.py's live roofline measurement) cost the
config-2 step -- steps with and without the profile, alternating; with the library selected by ICPFLOW_HIP_LIB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024, icp_max_iterations=50)
for _ in range(10): utils_match.hist_icp(a, s, d)
res = {"plain": [], "profiled": []}
for rnd in range(5):
    for name in ("plain", "profiled"):
        prof = _lib.Profile(64) if name == "profiled" else None
        torch.cuda.synchronize(); t = time.perf_counter()
        if prof is not None:
            with _lib.options(profile=prof):
                for _ in range(50): utils_match.hist_icp(a, s, d)
        else:
            for _ in range(50): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t) / 50 * 1e3)
        if prof is not None:
            icp, n = prof.collect(); prof.close(); last = icp / n
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), f"step plain {min(res['plain']):.4f} ms, with the events {min(res['profiled']):.4f} ms; ICP launch by the events {last:.4f} ms")
