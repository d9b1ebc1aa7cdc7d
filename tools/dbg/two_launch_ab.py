"""Developer tool: the ICP of batches larger than the GPU in ONE launch (the default; ICPFLOW_OPT_TWO_LAUNCH switches the two launches on) against two (all pairs to
iteration kSplitIter, then the pairs still moving; icp.hip icp_split_kernel) -- ICP launch(es) and step per shape, and that
transforms and iteration count are the same bits.  SHAPES="1024x2048,8192x2048" REPS=8 python tools/dbg/two_launch_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
shapes = os.environ.get("SHAPES", "1024x2048,8192x2048,2048x1024,600x2048,1500x1024,r2500x700,r900x2048,4000x200")
reps0 = int(os.environ.get("REPS", 8))
for sh in shapes.split(","):
    ragged = sh.startswith("r")
    B, N = map(int, sh.lstrip("r").split("x"))
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=30) if ragged else synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=int(os.environ.get("CAP", 50)))
    reps = max(2, reps0 * 1024 // max(B, 1024))
    res = {}
    for name, kw in (("one", dict()), ("two", dict(two_launch=True))):
        with _lib.options(**kw):
            T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
            torch.cuda.synchronize()
            prof = _lib.Profile(reps + 4)
            with _lib.options(profile=prof, **kw):
                t = time.perf_counter()
                for _ in range(reps):
                    utils_match.hist_icp(a, s, d)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t) / reps * 1e3
            icp, n = prof.collect(); prof.close()
        res[name] = (T, int(it), ms, icp / max(n, 1))
    same = torch.equal(res["one"][0], res["two"][0]) and res["one"][1] == res["two"][1]
    print(f"{sh}: one launch step {res['one'][2]:.3f} ms icp {res['one'][3]:.3f} | two launches step {res['two'][2]:.3f} ms icp {res['two'][3]:.3f} | "
          f"iterations {res['one'][1]} / {res['two'][1]} | same bits {same}", flush=True)
