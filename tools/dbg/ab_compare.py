"""Developer tool: time the registration path on a few shapes with the library selected by ICPFLOW_HIP_LIB
(run twice, once per library, on the same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match, frame_pairs, utils_track, utils_flow
dev = torch.device("cuda", 0)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
out = {}
for B, N, reps in ((256, 1024, 30), (1024, 2048, 8), (64, 1024, 30), (256, 512, 30)):
    S, D, _ = synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    prof = _lib.Profile(200)
    with _lib.options(profile=prof):
        ms = timeit(lambda: utils_match.hist_icp(a, s, d), reps)
    icp_ms, n = prof.collect()
    out[f"{B}x{N}"] = (round(ms, 4), round(icp_ms / n, 4))
g = np.load("tests/golden/g8_demo.npz"); lab = np.load("tests/golden/g8_demo_labels.npz")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd, ls, ld = G(g["point_src"]), G(g["point_dst"]), G(lab["label_src"]).float(), G(lab["label_dst"]).float()
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    def run():
        torch.manual_seed(0)
        pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
        return utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, torch.eye(4, device=dev))
    out[f"demo_mp{mp}"] = round(timeit(run, 5), 3)
print(os.environ.get("ICPFLOW_HIP_LIB", "default"), out)
