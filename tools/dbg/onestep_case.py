"""Developer tool: one (pair, step) of the teacher-forced test taken apart -- the oracle's correspondences, the HIP NN
primitive on the same moved points, the Kabsch step in numpy fp64 on those correspondences, HIP's one step under the
three searches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from icp_flow_amd import _lib, synthetic, utils_helper
from icp_flow_amd.utils_icp_pytorch3d import SimilarityTransform, iterative_closest_point
from oracle import reference_path as rp
import test_gpu_onestep as t1
S, D, _ = synthetic.make_batch(256, 1024, seed=0)
args = rp.default_args(max_points=1024, icp_max_iterations=50)
moved, fixed, sol = t1._oracle_trace(args, S, D, 50)
B = len(moved)
X0 = moved[:, :, :3]
dev = torch.device("cuda:0")
for k, b in ((4, 131), (2, 40), (15, 97)):
    Rk, Tk = sol.history[k - 1][0], sol.history[k - 1][1]
    Xt = torch.bmm(X0, Rk) + Tk[:, None, :]
    n_x = (moved[:, :, 3] > 0).sum(1); n_y = (fixed[:, :, 3] > 0).sum(1)
    d2, idx, nn = rp.knn_points(Xt, fixed[:, :, :3], n_x, n_y, return_nn=True)
    w = (moved[:, :, 3] > 0) & (d2 <= 0.1 ** 2)
    # HIP NN primitive on the oracle's Xt
    q = torch.cat([Xt, moved[:, :, 3:4]], 2).contiguous().to(dev)
    hi, hd = utils_helper.nearest_neighbor_batch(q, fixed.to(dev))
    same_idx = (hi.cpu()[b][: n_x[b]] == idx[b][: n_x[b]]).all().item()
    print(f"step {k} pair {b}: HIP nn primitive idx == oracle idx: {same_idx}; gated {int(w[b].sum())}")
    # Kabsch in fp64 numpy on the oracle's correspondences
    x = X0[b][w[b]].double().numpy(); y = nn[b][w[b]].double().numpy()
    mx, my = x.mean(0), y.mean(0)
    H = (x - mx).T @ (y - my) / len(x)
    U, Sg, Vt = np.linalg.svd(H)
    E = np.eye(3); E[2, 2] = np.linalg.det(U @ Vt)
    R = U @ E @ Vt
    T = my - mx @ R
    R1, T1 = sol.history[k][0][b].double().numpy(), sol.history[k][1][b].double().numpy()
    print(f"   numpy fp64 Kabsch on the oracle's correspondences vs the oracle's state k+1: |dR| {np.abs(R - R1).max():.2e} |dT| {np.abs(T - T1).max():.2e}; singular values {Sg}")
    if hasattr(_lib._L, "icpflow_debug_solve"):
        import ctypes
        _lib._L.icpflow_debug_solve(int(b), None)
        got = iterative_closest_point(moved.to(dev), fixed.to(dev), init_transform=SimilarityTransform(Rk.to(dev), Tk.to(dev), torch.ones(B, device=dev)), thres=0.1, max_iterations=2)
        torch.cuda.synchronize()
        buf = (ctypes.c_double * 64)()
        _lib._L.icpflow_debug_solve(0, buf)
        dbg = np.array(buf[:])
        o = dbg[45:48]
        xs, ys = x - o, y - o
        mom = np.concatenate([[len(x)], xs.sum(0), ys.sum(0), (xs[:, :, None] * ys[:, None, :]).sum(0).reshape(-1), [(xs ** 2).sum()], [(ys ** 2).sum()]])
        print("   kernel moments vs numpy (relative):", np.abs(dbg[:18] - mom) / np.maximum(np.abs(mom), 1e-30))
        print("   kernel H", dbg[26:35], "numpy H", H.reshape(-1), "max |dH|", np.abs(dbg[26:35] - H.reshape(-1)).max())
        print("   kernel lambda", dbg[35], "numpy s1+s2+s3(signed)", Sg[0] + Sg[1] + Sg[2] * np.sign(np.linalg.det(H)))
        print("   kernel R - numpy R", np.abs(dbg[36:45].reshape(3, 3) - R).max())
        xb = (ctypes.c_float * 12288)()
        _lib._L.icpflow_debug_xt(xb)
        kx = np.array(xb[:], dtype=np.float32).reshape(4096, 3)[: int(n_x[b])]
        ox = Xt[b][: int(n_x[b])].numpy()
        neq = (kx != ox).any(1)
        print(f"   kernel's moved points vs the oracle's Xt: {int(neq.sum())} of {len(ox)} rows differ; max |d| {np.abs(kx - ox).max():.3e}")
        # which FMA order does this host's torch.bmm use?
        xd, rd = X0[b].double().numpy(), Rk[b].double().numpy()
        f32 = lambda a: a.astype(np.float32).astype(np.float64)
        cand = {"fma(z,R2j,fma(y,R1j,x*R0j))": f32(f32(f32(xd[:, 0:1] * rd[0]) + xd[:, 1:2] * rd[1]) + xd[:, 2:3] * rd[2]),
                "no fma": f32(f32(f32(xd[:, 0:1] * rd[0]) + f32(xd[:, 1:2] * rd[1])) + f32(xd[:, 2:3] * rd[2])),
                "fma(x,R0j,fma(y,R1j,z*R2j))": f32(f32(f32(xd[:, 2:3] * rd[2]) + xd[:, 1:2] * rd[1]) + xd[:, 0:1] * rd[0])}
        bm = torch.bmm(X0[b:b + 1], Rk[b:b + 1])[0].numpy()
        print("   host torch.bmm vs candidate orders (rows differing):", {k: int((v.astype(np.float32) != bm).any(1).sum()) for k, v in cand.items()})
    for search in ("sweep", "scan", "grid"):
        with _lib.options(search=search):
            got = iterative_closest_point(moved.to(dev), fixed.to(dev), init_transform=SimilarityTransform(Rk.to(dev), Tk.to(dev), torch.ones(B, device=dev)), thres=0.1, max_iterations=2)
            rec = got.t_history.records()[0].cpu()
        Rg, Tg = rec[b, 0:9].reshape(3, 3).double().numpy(), rec[b, 9:12].double().numpy()
        print(f"   HIP {search}: count {int(rec[b, 14])}; vs numpy fp64 |dR| {np.abs(Rg - R).max():.2e} |dT| {np.abs(Tg - T).max():.2e}; vs oracle |dR| {np.abs(Rg - R1).max():.2e}")
    # per-pair run (batch of one): does the result depend on the batch?
    one = slice(b, b + 1)
    got = iterative_closest_point(moved[one].to(dev).contiguous(), fixed[one].to(dev).contiguous(), init_transform=SimilarityTransform(Rk[one].to(dev), Tk[one].to(dev), torch.ones(1, device=dev)), thres=0.1, max_iterations=2)
    rec = got.t_history.records()[0].cpu()
    print(f"   HIP alone (B = 1): count {int(rec[0, 14])}; vs numpy fp64 |dR| {np.abs(rec[0, 0:9].reshape(3, 3).double().numpy() - R).max():.2e}")
