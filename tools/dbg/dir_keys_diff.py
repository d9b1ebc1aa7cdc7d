import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from types import SimpleNamespace
from icp_flow_amd import _lib, synthetic, utils_match
dev = torch.device("cuda", 0)
for B, N, first in ((8192, 2048, 0), (4096, 1024, 20000), (2048, 1500, 40000)):
    S, D, _ = synthetic.make_batch(B, N, seed=0, first=first)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50)
    with _lib.options(no_dir_keys=True):
        T0, i0 = utils_match.hist_icp(a, s, d, return_iterations=True)
    T1, i1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    diff = (T0 != T1).flatten(1).any(1).cpu().numpy()
    p = S[:, :, :3].astype(np.float64)
    def mv(M): M = M.cpu().numpy().astype(np.float64); return np.einsum("bij,bnj->bni", M[:, :3, :3], p) + M[:, None, :3, 3]
    disp = np.abs(mv(T0) - mv(T1)).max(-1).max(-1)
    print(f"{B}x{N}: iterations {int(i0)} / {int(i1)}; pairs whose transform differs in any bit: {diff.sum()} of {B}; largest displacement of a point {disp.max():.3e} m; "
          f"pairs moved > 1e-6 m: {(disp > 1e-6).sum()}, > 1e-5: {(disp > 1e-5).sum()}, > 1e-4: {(disp > 1e-4).sum()}; worst {np.argsort(-disp)[:5].tolist()} {np.sort(disp)[-5:][::-1].round(7).tolist()}", flush=True)
