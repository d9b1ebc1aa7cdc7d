"""Developer tool: the config-2 step on torch's default stream (the null stream) against a created stream: ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
for B, N, reps in ((256, 1024, 50), (1024, 2048, 10)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    side = torch.cuda.Stream()
    ms = {"default": [], "created": []}
    for rnd in range(4):
        for name in ("default", "created"):
            ctx = torch.cuda.stream(side) if name == "created" else torch.cuda.stream(torch.cuda.default_stream())
            with ctx:
                for _ in range(5): utils_match.hist_icp(a, s, d)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(reps): utils_match.hist_icp(a, s, d)
                torch.cuda.synchronize(); ms[name].append((time.perf_counter() - t) / reps * 1e3)
    print(f"{B} x {N}: default stream {min(ms['default']):.4f} ms per step, created stream {min(ms['created']):.4f} ms", flush=True)
