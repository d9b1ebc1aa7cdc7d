"""Developer tool: the sweeps' sort keys -- the fixed cloud's longest axis (ICPFLOW_OPT_NO_DIR_KEYS) against the best of three axes and
six horizontal directions (csrc/sortdir.hpp): step and ICP launch per shape, which keys the pairs got, and that the transforms are the
same bits.  SHAPES="1024x2048,256x1024" python tools/dbg/dir_keys_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
shapes = os.environ.get("SHAPES", "256x1024,1024x2048,600x2048,1500x1500,r900x2048,m128x4000,8192x2048")
reps0 = int(os.environ.get("REPS", 8))
for sh in shapes.split(","):
    ragged = True if sh.startswith("r") else ("matched" if sh.startswith("m") else False)
    B, N = map(int, sh.lstrip("rm").split("x"))
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=ragged, n_min=30) if ragged else synthetic.make_batch(B, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=int(os.environ.get("CAP", 50)))
    reps = max(2, reps0 * 1024 // max(B, 1024))
    res = {}
    for name, kw in (("axis", dict(no_dir_keys=True)), ("dir", dict())):
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
    same = torch.equal(res["axis"][0], res["dir"][0]) and res["axis"][1] == res["dir"][1]
    print(f"{sh}: axis keys step {res['axis'][2]:.3f} ms icp {res['axis'][3]:.3f} | direction keys step {res['dir'][2]:.3f} ms icp {res['dir'][3]:.3f} | "
          f"iterations {res['axis'][1]} / {res['dir'][1]} | same bits {same}", flush=True)
