"""Developer tool: the config-2 step (and config 4's shard) launched call by call against the same call captured into a HIP graph
(torch.cuda.CUDAGraph) and replayed: ms per step, results compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
for B, N, reps in ((256, 1024, 50), (1024, 2048, 10), (64, 512, 50)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    want = utils_match.hist_icp(a, s, d).clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): utils_match.hist_icp(a, s, d)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = utils_match.hist_icp(a, s, d)
    out.zero_(); g.replay(); torch.cuda.synchronize()
    same = torch.equal(out, want)
    ms = {"calls": [], "graph": []}
    for rnd in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): utils_match.hist_icp(a, s, d)
        torch.cuda.synchronize(); ms["calls"].append((time.perf_counter() - t) / reps * 1e3)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): g.replay()
        torch.cuda.synchronize(); ms["graph"].append((time.perf_counter() - t) / reps * 1e3)
    print(f"{B} x {N}: call by call {min(ms['calls']):.4f} ms per step, graph replay {min(ms['graph']):.4f} ms; identical {same}", flush=True)
