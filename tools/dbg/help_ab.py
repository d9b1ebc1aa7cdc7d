"""Developer tool: the persistent ICP launch with and without helpers (results compared bit for bit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out
shapes = ((1024, 2048, False, 6), (600, 2048, False, 6), (2048, 2048, False, 4), (8192, 2048, False, 2), (700, 3000, True, 6), (1024, 1500, False, 6))
if os.environ.get("QUICK"): shapes = shapes[:2]
for B, N, ragged, reps in shapes:
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=300) if ragged else synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    res = {}
    for tag, kw in (("helpers", {}), ("no helpers", {"no_helpers": True}), ("hardware dispatch", {"no_persistent": True})):
        prof = _lib.Profile(64)
        with _lib.options(profile=prof, **kw):
            ms, (T, it) = timeit(lambda: utils_match.hist_icp(a, s, d, return_iterations=True), reps)
        icp_ms, n = prof.collect()
        res[tag] = (ms, icp_ms / n, T, int(it))
    same = torch.equal(res["helpers"][2], res["no helpers"][2]) and torch.equal(res["helpers"][2], res["hardware dispatch"][2])
    print(f"{B} x {N}{' ragged' if ragged else ''}: icp " + ", ".join(f"{k} {v[1]:.3f} ms" for k, v in res.items()) +
          f"; step {res['helpers'][0]:.3f} vs {res['no helpers'][0]:.3f} ms; iterations {res['helpers'][3]}; identical: {same}; finite: {bool(torch.isfinite(res['helpers'][2]).all())}", flush=True)
