"""Developer tool: ICP launch and step time with the ticket-dispatched persistent grid against one workgroup per pair
dealt by the hardware (ICPFLOW_OPT_NO_PERSISTENT), results compared bit for bit."""
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
for B, N, ragged, reps in ((1024, 2048, False, 8), (2048, 2048, False, 5), (8192, 2048, False, 3), (600, 1024, True, 10), (2048, 1024, False, 8), (1024, 1500, False, 8), (4096, 512, True, 8), (300, 10000, True, 3)):
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=40) if ragged else synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    res = {}
    for tag, kw in (("ticket", {}), ("hardware", {"no_persistent": True})):
        prof = _lib.Profile(64)
        with _lib.options(profile=prof, **kw):
            ms, T = timeit(lambda: utils_match.hist_icp(a, s, d), reps)
        icp_ms, n = prof.collect()
        res[tag] = (ms, icp_ms / n, T)
    print(f"{B} x {N}{' ragged' if ragged else ''}: step {res['ticket'][0]:.3f} ms (icp {res['ticket'][1]:.3f}) vs hardware dispatch {res['hardware'][0]:.3f} ms (icp {res['hardware'][1]:.3f}); "
          f"{B / res['ticket'][0]:.1f} vs {B / res['hardware'][0]:.1f} k registrations/s; identical: {torch.equal(res['ticket'][2], res['hardware'][2])}")
