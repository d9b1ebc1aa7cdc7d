"""Developer tool: the scoring sweeps with the occupancy pre-bound (default) against ICPFLOW_OPT_NO_SCORE_PREBOUND, alternating in one
process: transforms bit for bit, step time of a few shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
shapes = ((256, 1024, 0, False, 40), (1024, 2048, 0, False, 8), (600, 1024, 31, True, 20), (128, 10000, 0, True, 10), (8192, 2048, 0, False, 2))
if os.environ.get("QUICK"): shapes = shapes[:4]
for B, N, seed, ragged, reps in shapes:
    S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=20) if ragged else synthetic.make_batch(B, N, seed=seed)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    T1 = utils_match.hist_icp(a, s, d)
    with _lib.options(no_score_prebound=True):
        T0 = utils_match.hist_icp(a, s, d)
    same = torch.equal(T0, T1)
    ms = {}
    for rnd in range(3):
        for name, opts in (("prebound", {}), ("plain", {"no_score_prebound": True})):
            with _lib.options(**opts):
                utils_match.hist_icp(a, s, d); torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps): utils_match.hist_icp(a, s, d)
                torch.cuda.synchronize()
                ms.setdefault(name, []).append((time.perf_counter() - t) / reps * 1e3)
    print(f"{B} x {N}{' ragged' if ragged else ''}: identical {same}; step plain {min(ms['plain']):.4f} ms -> pre-bound {min(ms['prebound']):.4f} ms", flush=True)
