"""Developer check: throughput of BASELINE config 2 when consecutive (independent) batches alternate between
S HIP streams, against the single-stream step of bench.py."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from icp_flow_amd import synthetic, utils_match
B, N = 256, 1024
dev = torch.device("cuda", 0)
args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50,
                       icp_stop_mode="reference")
for S in (1, 2, 3, 4):
    batches = []
    for k in range(S):
        s_, d_, _ = synthetic.make_batch(B, N, seed=0, first=0)
        batches.append((torch.from_numpy(s_).to(dev), torch.from_numpy(d_).to(dev)))
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    def run(steps):
        outs = []
        for i in range(steps):
            with torch.cuda.stream(streams[i % S]):
                outs.append(utils_match.hist_icp(args, *batches[i % S]))
        return outs
    run(2 * S); torch.cuda.synchronize()
    t0 = time.perf_counter(); outs = run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0], o) for o in outs)
    print(f"{S} stream(s): {B * 40 / dt:10.0f} registrations/s, {dt / 40 * 1e3:.3f} ms per batch, identical results {same}")
