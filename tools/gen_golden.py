#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own vendored Python on
CPU tensors.  Runs ONLY in the build container (needs /root/reference); the GPU
box never sees the reference -- only the arrays written here travel.

What executes unmodified from /root/reference: utils_match.{hist_icp,match_eval},
utils_hist.{topk_nms,estimate_init_pose}, utils_icp.{apply_icp,pytorch3d_icp},
utils_icp_pytorch3d.{iterative_closest_point,corresponding_points_alignment},
utils_helper.{nearest_neighbor_batch,transform_points_batch,pad_segment}.
What is a stand-in (tools/standins/, this repo's own code): the un-installable
third-party modules -- pytorch3d's knn_points/wmean/matrix_to_euler_angles and the
CUDA-only hist_cuda vote, both backed by oracle/oracle_core.c.

Fixtures hold inputs and expected outputs only (no reference source text).

Usage:  python tools/gen_golden.py [--only g1,g5] [--demo]
"""
import argparse
import os
import sys
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tools", "standins"))

import matplotlib  # noqa: E402

matplotlib.use("Agg")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import utils_helper  # noqa: E402  (reference)
import utils_hist  # noqa: E402  (reference)
import utils_icp  # noqa: E402  (reference)
import utils_icp_pytorch3d  # noqa: E402  (reference)
import utils_match  # noqa: E402  (reference)
from hist_cuda.hist import hist as ref_hist  # noqa: E402  (stand-in vote)

from icp_flow_amd import synthetic  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def args_ns(**kw):
    a = dict(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=256,
             thres_iou=0.2, thres_rot=0.1, thres_error=0.2, thres_box=0.1, min_cluster_size=20)
    a.update(kw)
    return SimpleNamespace(**a)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    meta = dict(torch_version=np.array(torch.__version__), generator=np.array("tools/gen_golden.py"))
    np.savez_compressed(path, **arrays, **meta)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def sparse(bins):
    flat = bins.reshape(-1)
    nz = np.flatnonzero(flat)
    return nz.astype(np.int64), flat[nz].astype(np.float32)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def ragged_batch(B, N, seed, first=0, n_min=20):
    S, D, T = synthetic.make_batch(B, N, seed=seed, first=first, ragged=True, n_min=n_min)
    return t(S), t(D), T



def cut_is_tied(a, src, dst, swap_smaller_first=False, any_votes=False):
    """True where the 5th and 6th surviving peak share a POSITIVE vote count: which peaks make
    the top-5 then depends on torch.topk's implementation-defined tie order, so
    the chosen initial pose of that pair is not a portable expectation."""
    if swap_smaller_first:
        n1 = (src[:, :, 3] > 0).sum(1)
        n2 = (dst[:, :, 3] > 0).sum(1)
        sw = n1 > n2
        s2, d2 = src.clone(), dst.clone()
        s2[sw] = dst[sw]
        d2[sw] = src[sw]
        src, dst = s2, d2
    tf, th = a.translation_frame, a.thres_dist
    ex = torch.arange(-tf, tf + th - 1e-8, th)
    ez = torch.arange(-th, 2 * th - 1e-8, th)
    h = ref_hist(dst, src, ex.min(), ex.min(), ez.min(), ex.max(), ex.max(), ez.max(),
                 len(ex), len(ex), len(ez))
    xp = torch.nn.functional.max_pool3d(h[:, None], kernel_size=11, stride=1, padding=5)
    surv = (h[:, None] * (h[:, None] == xp).float()).reshape(len(h), -1)
    top = torch.sort(surv, dim=1, descending=True)[0][:, :6]
    if any_votes:   # fewer than five positive peaks: zero-vote bins complete the top-5 in implementation-defined order
        return (top[:, 4] == top[:, 5]).numpy()
    return ((top[:, 4] == top[:, 5]) & (top[:, 4] > 0)).numpy()


# --------------------------------------------------------------------------
def g1_hist():
    # (a) the reference's only known-answer script, hist_cuda/test.py:16-50
    torch.manual_seed(2022)
    pts = torch.randn(3, 1000, 3)
    ind = torch.randint(0, 2, size=(3, 1000, 1))
    p1 = torch.cat([pts, ind], dim=-1)
    p2 = p1.clone()
    p2[:, :, 0] += 5.0
    p2[:, :, 1] += -3.0
    p2[:, :, 2] += -0.2
    bx = torch.arange(-10.0, 10.0 + 0.1, 0.1)
    bz = torch.arange(-0.5, 0.5 + 0.1, 0.1)
    h = ref_hist(p1, p2, -10.0, -10.0, -0.5, 10.0, 10.0, 0.5, len(bx), len(bx), len(bz)).numpy()
    nz, val = sparse(h)
    save("g1_hist_testpy", X=p1.numpy(), Y=p2.numpy(),
         mins=np.array([-10.0, -10.0, -0.5], np.float32), maxs=np.array([10.0, 10.0, 0.5], np.float32),
         lens=np.array([len(bx), len(bx), len(bz)], np.int64), bins_shape=np.array(h.shape),
         bins_nz=nz, bins_val=val,
         argmax=np.array([int(x.argmax()) for x in h]), peak=np.array([float(x.max()) for x in h]),
         total=np.array([float(x.sum()) for x in h]))

    # (b) reference-style calls (utils_hist.py:61-72): X=dst, Y=src, min/max from arange edges
    for tag, tf in (("tf2p0", 2.0), ("tf3p34", 3.34)):
        a = args_ns(translation_frame=tf)
        src, dst, _ = ragged_batch(6, 192, seed=11)
        # pair 5: hand-made differences exactly on bin edges / box borders / outside
        ex = torch.arange(-tf, tf + a.thres_dist - 1e-8, a.thres_dist)
        ez = torch.arange(-a.thres_dist, 2 * a.thres_dist - 1e-8, a.thres_dist)
        src[5] = 1e8
        src[5, :, 3] = 0
        dst[5] = 1e8
        dst[5, :, 3] = 0
        base = torch.tensor([10.0, -20.0, 0.5])
        src[5, 0, 0:3] = base
        src[5, 0, 3] = 1
        deltas = [
            (float(ex.min()), 0.0, 0.0), (float(ex.max()), 0.0, 0.0),          # on min (in) / on max (out)
            (0.0, float(ex.min()), float(ez.min())), (0.0, 0.0, float(ez.max())),
            (float(ex[3]), float(ex[7]), 0.0), (float(ex[-2]), float(ex[1]), float(ez[1])),
            (np.nextafter(np.float32(ex.max()), np.float32(0)).item(), 0.05, -0.05),
            (5.0 * tf, 0.0, 0.0), (0.0, 0.0, 0.25), (0.3, -0.7, 0.02), (0.3, -0.7, 0.02),
        ]
        for i, d in enumerate(deltas):
            dst[5, i, 0:3] = base + torch.tensor(d, dtype=torch.float32)
            dst[5, i, 3] = 1
        h = ref_hist(dst, src, ex.min(), ex.min(), ez.min(), ex.max(), ex.max(), ez.max(),
                     len(ex), len(ex), len(ez)).numpy()
        nz, val = sparse(h)
        votes, idx = utils_hist.topk_nms(torch.from_numpy(h))
        save("g1_hist_ref_" + tag, src=src.numpy(), dst=dst.numpy(),
             translation_frame=np.array(tf), thres_dist=np.array(a.thres_dist),
             mins=np.array([float(ex.min()), float(ex.min()), float(ez.min())], np.float32),
             maxs=np.array([float(ex.max()), float(ex.max()), float(ez.max())], np.float32),
             lens=np.array([len(ex), len(ex), len(ez)], np.int64),
             edges_x=ex.numpy(), edges_z=ez.numpy(), bins_shape=np.array(h.shape),
             bins_nz=nz, bins_val=val,
             # G2: reference topk_nms on these bins (tie order inside torch.topk is
             # implementation-defined; tests compare tie-free prefixes only)
             peak_votes=votes.numpy(), peak_idx=idx.numpy())


def g3_nn():
    src, dst, _ = ragged_batch(4, 160, seed=23)
    idx, dist = utils_helper.nearest_neighbor_batch(src, dst)
    idx2, dist2 = utils_helper.nearest_neighbor_batch(dst, src)
    save("g3_nn", src=src.numpy(), dst=dst.numpy(), idx_fwd=idx.numpy(), dist_fwd=dist.numpy(),
         idx_bwd=idx2.numpy(), dist_bwd=dist2.numpy())


def g4_init_pose():
    a = args_ns(chunk_size=3)
    src, dst, Tt = ragged_batch(8, 384, seed=31, n_min=90)
    T = utils_hist.estimate_init_pose(a, src, dst)
    save("g4_init_pose", src=src.numpy(), dst=dst.numpy(), T_true=Tt, T_init=T.numpy(),
         cut_tied=cut_is_tied(a, src, dst),
         translation_frame=np.array(a.translation_frame), thres_dist=np.array(a.thres_dist))


def _icp_case(X, Y, thres=0.1):
    sol = utils_icp_pytorch3d.iterative_closest_point(X, Y, thres=thres, max_iterations=100,
                                                      relative_rmse_thr=1e-6)
    hist_R = np.stack([h.R.numpy() for h in sol.t_history])
    hist_T = np.stack([h.T.numpy() for h in sol.t_history])
    return dict(R=sol.RTs.R.numpy(), T=sol.RTs.T.numpy(), rmse=sol.rmse.numpy(),
                converged=np.array(bool(sol.converged)), iterations=np.array(len(sol.t_history)),
                hist_R=hist_R, hist_T=hist_T, Xt=sol.Xt.numpy())


def g5_icp():
    out = {}
    cases = {}

    def prealigned(B, N, seed, ragged, rot_deg, shift):
        """src moved by the ground-truth motion, then perturbed by a small yaw about
        its centroid and a sub-threshold shift, so that ICP starts with inliers."""
        S, D, Tt = synthetic.make_batch(B, N, seed=seed, ragged=ragged)
        src, dst = t(S), t(D)
        for i in range(B):
            v = src[i, :, 3] > 0
            Ti = torch.from_numpy(Tt[i])
            p = src[i, v, 0:3] @ Ti[:3, :3].T + Ti[:3, 3]
            ang = np.deg2rad(rot_deg * (1.0 + 0.15 * i) * (-1) ** i)
            c, s_ = np.cos(ang), np.sin(ang)
            Rz = torch.tensor([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=torch.float32)
            ctr = p.mean(0)
            src[i, v, 0:3] = (p - ctr) @ Rz.T + ctr + torch.tensor(shift, dtype=torch.float32)
        return src, dst

    cases["a"] = prealigned(1, 64, 41, False, 1.0, [0.03, -0.02, 0.01])
    cases["b"] = prealigned(4, 256, 43, True, 1.5, [0.04, 0.03, -0.02])
    cases["c"] = prealigned(16, 128, 47, False, 0.8, [0.02, 0.015, -0.01])
    # case d: degenerate members -- pair 0 regular, pair 1 ZERO inliers (10 m apart: the
    # batch can then never satisfy the all-pairs stop), pair 2 planar (z const)
    src, dst = prealigned(3, 96, 53, False, 0.5, [0.02, -0.01, 0.0])
    dst[1, :, 0] += 10.0
    src[2, :, 2] = 0.7
    dst[2, :, 2] = 0.7
    cases["d"] = (src, dst)
    # case e: B=6 ragged, larger perturbation (more iterations)
    cases["e"] = prealigned(6, 384, 59, True, 2.5, [0.06, -0.05, 0.02])
    from oracle import reference_path as rp
    for k, (src, dst) in cases.items():
        res = _icp_case(src, dst)
        tr = rp.iterative_closest_point(src, dst, trace=True)
        assert tr.iterations == int(res["iterations"])
        # fewer than 3 gated correspondences => rank-deficient covariance => the rotation is
        # not unique and depends on the SVD backend: not a portable expectation
        out[f"{k}_min_inliers"] = torch.stack([h[3] for h in tr.history]).min(0)[0].numpy()
        out[f"{k}_src"] = src.numpy()
        out[f"{k}_dst"] = dst.numpy()
        for kk, vv in res.items():
            out[f"{k}_{kk}"] = vv
        print(f"  icp case {k}: iterations={int(res['iterations'])} converged={bool(res['converged'])}")
    save("g5_icp", **out)


def g6_hist_icp():
    a = args_ns(chunk_size=4)
    # ragged batch: some pairs have n_src > n_dst (swapped inside hist_icp)
    src, dst, Tt = ragged_batch(10, 384, seed=61, n_min=90)
    ns = (src[:, :, 3] > 0).sum(1)
    nd = (dst[:, :, 3] > 0).sum(1)
    T = utils_match.hist_icp(a, src, dst)
    # also the two stages separately on the un-swapped batch
    init = utils_hist.estimate_init_pose(a, src, dst)
    Ti = utils_icp.apply_icp(a, src, dst, init.clone())
    ev = utils_match.match_eval(a, src, dst, T)
    from oracle import reference_path as rp
    _, aux = rp.apply_icp(a, src, dst, init.clone(), return_aux=True)
    save("g6_hist_icp", src=src.numpy(), dst=dst.numpy(), T_true=Tt, n_src=ns.numpy(), n_dst=nd.numpy(),
         T_hist_icp=T.numpy(), T_init_noswap=init.numpy(), T_apply_icp_noswap=Ti.numpy(),
         cut_tied=cut_is_tied(a, src, dst, True), cut_tied_noswap=cut_is_tied(a, src, dst),
         rolled_back_noswap=aux["rolled_back"].numpy(), icp_iterations_noswap=np.array(aux["iterations"]),
         translation_frame=np.array(a.translation_frame), thres_dist=np.array(a.thres_dist),
         # G7: match_eval on the final transforms
         ev_errors=ev[0].numpy(), ev_inliers=ev[1].numpy(), ev_ratios=ev[2].numpy(),
         ev_ious=ev[3].numpy(), ev_translations=ev[4].numpy(), ev_rotations=ev[5].numpy())

    # roll-back: identical clouds => the (zero) init pose is already perfect, e_icp >= e_init.
    # Kept apart from the batch above: with exact duplicates the rmse is pure rounding noise
    # (or exactly 0 -> NaN relative rmse), so the batch-global iteration count of a batch that
    # contains such a pair is an fp accident rather than a portable expectation.
    S, D, _ = synthetic.make_batch(2, 128, seed=71, ragged=True)
    src, dst = t(S), t(D)
    dst[:] = src
    init = utils_hist.estimate_init_pose(a, src, dst)
    T = utils_match.hist_icp(a, src, dst)
    save("g6_rollback", src=src.numpy(), dst=dst.numpy(), T_init=init.numpy(), T_hist_icp=T.numpy(),
         translation_frame=np.array(a.translation_frame))

    # dense (config-2 shaped, scaled down) batch: n = N, 8 pairs x 512 points
    S, D, Tt = synthetic.make_batch(8, 512, seed=67)
    src, dst = t(S), t(D)
    a = args_ns(max_points=512)
    T = utils_match.hist_icp(a, src, dst)
    ev = utils_match.match_eval(a, src, dst, T)
    save("g6_hist_icp_dense", seed=np.array(67), num_pairs=np.array(8), max_points=np.array(512),
         T_true=Tt, T_hist_icp=T.numpy(), cut_tied=cut_is_tied(a, src, dst, True),
         ev_errors=ev[0].numpy(), ev_inliers=ev[1].numpy(), ev_ratios=ev[2].numpy(),
         ev_ious=ev[3].numpy(), ev_translations=ev[4].numpy(), ev_rotations=ev[5].numpy())


def topk_cuda_order(x, k, dim=1, **kw):
    """torch.topk with the tie order of ATen's CUDA implementation (what the reference runs on): the single-block
    radix select (aten/src/ATen/native/cuda/TensorTopK.cu, gatherTopK -- slices of 41*41*3 = 5043 elements stay far
    below the multi-block thresholds) first finds the k-th value by radix selection, then writes every element
    strictly greater than it and completes the k outputs with elements EQUAL to it in ascending index order (an
    exclusive prefix scan over the slice in index order).  A stable descending sort returns exactly that set; torch-CPU
    (std::partial_sort on (value, index) pairs) picks another subset of the tied elements."""
    order = torch.argsort(x, dim=dim, descending=True, stable=True).narrow(dim, 0, k)
    return torch.gather(x, dim, order), order


def g8_demo(max_points=2048, name="g8_demo", topk_rule="torch_cpu"):
    """topk_rule "cuda": torch.topk replaced by `topk_cuda_order` while the reference's code runs (everything else
    unmodified) -- the run the reference would make on its own platform as far as the tie order goes.

    BASELINE config 1: the reference's demo frame pair (demo.npz) through the reference's own
    match_pcds (both association stages) and flow_estimation_torch, on CPU.  Cluster labels come
    from sklearn's HDBSCAN through the reference's cluster_pcd (the pinned `hdbscan` package is
    not installable), so the labels are part of the fixture.  max_points = 2048 keeps the
    reference's padded N^2 scans tractable on CPU (SURVEY A.8 probe 3); `g8mp10000` is the same run at the
    reference's real setting, max_points = 10000 (demo.sh:9-13): 16 padded 10^4 x 10^4 scans per pair through
    oracle_core.c -- the over-long clusters are subsampled with the reference's own torch.randperm stream
    (seed 0, main.py:139)."""
    import utils_flow  # noqa: E402  (reference)
    data = np.load(os.path.join(REF, "demo.npz"))
    src = data["pc1"][data["pc1_flows_valid_idx"]].astype(np.float32)      # demo.py:37-51
    dst = data["pc2"][data["pc2_flows_valid_idx"]].astype(np.float32)
    gt = data["gt_flow_0_1"][data["pc1_flows_valid_idx"]].astype(np.float32)
    a = args_ns(max_points=max_points, min_cluster_size=20, num_clusters=200, epsilon=0.25, if_hdbscan=True,
                translation_frame=2.0, thres_dist=0.1, thres_box=0.1, thres_rot=0.1, thres_error=0.2,
                thres_iou=0.2, chunk_size=50)
    lab_path = os.path.join(OUT, "g8_demo_labels.npz")
    if os.path.exists(lab_path):
        lab = np.load(lab_path)
        label_src, label_dst = lab["label_src"], lab["label_dst"]
    else:
        import utils_cluster  # noqa: E402  (reference; hdbscan stand-in = sklearn)
        labels = utils_cluster.cluster_pcd(a, np.concatenate([dst, src], axis=0),
                                           np.ones(len(src) + len(dst)).astype(bool))   # demo.py:210
        label_src = labels[len(dst):].astype(np.float32)
        label_dst = labels[0:len(dst)].astype(np.float32)
        np.savez_compressed(lab_path, label_src=label_src, label_dst=label_dst)
    torch.manual_seed(0)                                                         # main.py:139
    ps, pd = torch.from_numpy(src), torch.from_numpy(dst)
    ls, ld = torch.from_numpy(label_src).float(), torch.from_numpy(label_dst).float()
    # Per association stage, what the reference's own functions return on the way (observers around its unmodified
    # code): the candidate pairs, the initial poses of estimate_init_pose (smaller cloud first), whether the cut
    # between the 5th and 6th surviving peak is tied (torch.topk's order among equal votes is implementation-defined:
    # the pose of such a pair -- and, through the batch-global stop, the iteration count of its whole batch -- is not
    # a portable expectation), the ICP iteration count and hist_icp's transforms.
    stages = []
    orig = dict(pairs=utils_match.match_pairs, init=utils_match.estimate_init_pose, icp=utils_icp.iterative_closest_point,
                hist_icp=utils_match.hist_icp)

    def match_pairs_obs(args, sp, dp, sl, dl, prs):
        stages.append(dict(pairs=prs.numpy().astype(np.float32)))
        return orig["pairs"](args, sp, dp, sl, dl, prs)

    def init_obs(args, s_, d_):
        out = orig["init"](args, s_, d_)
        stages[-1]["init"] = out.numpy().copy()
        stages[-1]["tied"] = cut_is_tied(args, s_, d_, any_votes=True)
        return out

    def icp_obs(*x, **kw):
        sol = orig["icp"](*x, **kw)
        stages[-1]["iterations"] = len(sol.t_history)
        stages[-1]["converged"] = bool(sol.converged)
        return sol

    def hist_icp_obs(args, s_, d_):
        out = orig["hist_icp"](args, s_, d_)
        stages[-1]["T"] = out.numpy().copy()
        return out

    utils_match.match_pairs, utils_match.estimate_init_pose = match_pairs_obs, init_obs
    utils_icp.iterative_closest_point, utils_match.hist_icp = icp_obs, hist_icp_obs
    torch_topk = torch.topk
    if topk_rule == "cuda":
        torch.topk = topk_cuda_order          # (utils_hist.py:27 is the path's only torch.topk call)
    try:
        pairs, T = utils_match.match_pcds(a, ps, pd, ls, ld)
    finally:
        torch.topk = torch_topk
        utils_match.match_pairs, utils_match.estimate_init_pose = orig["pairs"], orig["init"]
        utils_icp.iterative_closest_point, utils_match.hist_icp = orig["icp"], orig["hist_icp"]
    stage_arrays = dict(stage_sizes=np.array([len(st["pairs"]) for st in stages]),
                        stage_pairs=np.concatenate([st["pairs"] for st in stages]),
                        stage_init=np.concatenate([st["init"] for st in stages]),
                        stage_tied=np.concatenate([st["tied"] for st in stages]),
                        stage_T=np.concatenate([st["T"] for st in stages]),
                        stage_iterations=np.array([st["iterations"] for st in stages]),
                        stage_converged=np.array([st["converged"] for st in stages]))
    print("  stages:", [(len(st["pairs"]), st["iterations"], st["converged"], int(st["tied"].sum())) for st in stages])
    flow = utils_flow.flow_estimation_torch(a, src_points=ps, dst_points=pd, src_labels=ls, dst_labels=ld,
                                            pairs=pairs, transformations=T, pose=torch.eye(4))
    epe = float(np.linalg.norm(flow.numpy() - gt, axis=1).mean())
    print(f"  demo: {len(pairs)} matched pairs, EPE vs gt {epe:.4f} m (zero flow {np.linalg.norm(gt, axis=1).mean():.4f})")
    inputs = dict(point_src=src, point_dst=dst, gt_flow=gt) if name == "g8_demo" else {}    # the inputs live in g8_demo.npz
    save(name, **inputs, **stage_arrays, pairs=pairs.numpy(), transformations=T.numpy(),
         flow=flow.numpy(), max_points=np.array(a.max_points), epe=np.array(epe))


def g9_epe():
    """Scene-flow metrics of the accuracy harness: the reference's own
    compute_epe_test / AverageMeter / average_meter (utils_eval.py:65-182) on the G8 flow vs the demo
    frame's ground truth, unmasked, masked, and accumulated over two 'frames' (the two halves)."""
    import utils_eval  # noqa: E402  (reference)
    g = np.load(os.path.join(OUT, "g8_demo.npz"))
    flow, gt = g["flow"], g["gt_flow"]
    rng = np.random.default_rng(9)
    mask = (rng.random(len(flow)) < 0.37).astype(np.float32)
    whole = np.array(utils_eval.compute_epe_test(flow, gt), dtype=np.float64)
    masked = np.array(utils_eval.compute_epe_test(flow, gt, mask), dtype=np.float64)
    h = len(flow) // 3
    meter = utils_eval.AverageMeter()
    per_frame = []
    for sl in (slice(0, h), slice(h, len(flow))):
        m = utils_eval.compute_epe_test(flow[sl], gt[sl])
        per_frame.append(m)
        meter.update(*m, sl.stop - sl.start)
    avg = np.array([meter.epe_avg, meter.accs_avg, meter.accr_avg, meter.outlier_avg, meter.Routlier_avg], dtype=np.float64)
    am = utils_eval.average_meter([m[0] for m in per_frame], [h, len(flow) - h])
    # a tiny hand case incl. zero ground-truth flow (relative error against 1e-20)
    tg = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 0.1], [3, 4, 0]], dtype=np.float32)
    tp = np.array([[0, 0, 0], [1.04, 0, 0], [0, 2.5, 0], [0, 0, 0.16], [3, 4, 0.31]], dtype=np.float32)
    tiny = np.array(utils_eval.compute_epe_test(tp, tg), dtype=np.float64)
    save("g9_epe", mask=mask, whole=whole, masked=masked, split=np.array(h), per_frame=np.array(per_frame, dtype=np.float64),
         meter_avg=avg, average_meter_epe=np.array(am), tiny_gt=tg, tiny_pred=tp, tiny=tiny)
    print("  epe/accs/accr/outlier/Routlier:", whole)


def g10_dbscan():
    """the reference's cluster_dbscan / cluster_pcd (utils_cluster.py:32-63,
    if_hdbscan False) on (a) the demo frame pair stacked as demo.py:210 does and (b) small synthetic
    clouds incl. a ground mask, with open3d's cluster_dbscan backed by sklearn (tools/standins/open3d).
    Also records that no point pair sits at distance == eps (where sklearn and nanoflann differ)."""
    import utils_cluster  # noqa: E402  (reference)
    from oracle import cluster as oc
    data = np.load(os.path.join(REF, "demo.npz"))
    src = data["pc1"][data["pc1_flows_valid_idx"]].astype(np.float32)
    dst = data["pc2"][data["pc2_flows_valid_idx"]].astype(np.float32)
    pts = np.concatenate([dst, src], axis=0)                                     # demo.py:210
    out = {}

    def ties(p, eps):
        """pairs at exactly eps (excluded by nanoflann's strict test) and pairs between the stand-in's
        radius and eps (must be none for the stand-in to equal the strict test)"""
        P = p[:, :3].astype(np.float64)
        from scipy.spatial import cKDTree
        c = cKDTree(P).query_pairs(eps * (1 + 1e-9), output_type="ndarray")
        d = P[c[:, 0]] - P[c[:, 1]]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        r = np.nextafter(eps, 0.0)
        assert np.count_nonzero((d2 > r * r) & (d2 < eps * eps)) == 0
        return int(np.count_nonzero(d2 == eps * eps))

    for tag, (eps, mcs, ncl) in dict(a=(0.25, 20, 200), b=(0.25, 30, 100), c=(0.4, 10, 50)).items():
        a = args_ns(epsilon=eps, min_cluster_size=mcs, num_clusters=ncl, if_hdbscan=False)
        lab = utils_cluster.cluster_pcd(a, pts, np.ones(len(pts)).astype(bool))
        mine = oc.cluster_pcd(a, pts, np.ones(len(pts)).astype(bool))
        out[f"demo_{tag}_pairs_at_eps"] = np.array(ties(pts, eps))
        print(f"  demo dbscan {tag}: eps {eps} min {mcs} keep {ncl}: {len(np.unique(lab)) - 1} clusters kept, "
              f"{int((lab == -1).sum())} unclustered, oracle mismatches {int((lab != mine).sum())}")
        out[f"demo_{tag}_params"] = np.array([eps, mcs, ncl], dtype=np.float64)
        out[f"demo_{tag}_labels"] = lab.astype(np.int32)
    # small clouds: blobs + bridges + noise, a ground mask (cluster_pcd's idxs_nonground)
    rng = np.random.default_rng(10)
    for k, n in enumerate((300, 2000, 9000)):
        centers = rng.uniform(-8, 8, size=(12, 3)) * np.array([1, 1, 0.2])
        p = centers[rng.integers(0, 12, n)] + rng.normal(0, 0.25, size=(n, 3)) * np.array([1, 1, 0.5])
        p[: n // 6] = rng.uniform(-10, 10, size=(n // 6, 3)) * np.array([1, 1, 0.2])      # clutter
        p = p.astype(np.float32)
        nonground = p[:, 2] > -0.5
        a = args_ns(epsilon=0.3, min_cluster_size=6, num_clusters=6, if_hdbscan=False)
        lab = utils_cluster.cluster_pcd(a, p, nonground)
        ties(p[nonground], 0.3)
        mine = oc.cluster_pcd(a, p, nonground)
        lit = oc.cluster_pcd(a, p, nonground, impl=oc.dbscan_index_order)
        print(f"  small {k}: n {n}: kept {len(np.unique(lab[lab >= 0]))} clusters; oracle mismatches "
              f"{int((lab != mine).sum())} / {int((lab != lit).sum())}")
        out[f"small_{k}_points"] = p
        out[f"small_{k}_nonground"] = nonground
        out[f"small_{k}_params"] = np.array([0.3, 6, 6], dtype=np.float64)
        out[f"small_{k}_labels"] = lab
    save("g10_dbscan", **out)


def g11_hdbscan():
    """SURVEY 8(f) row 4, HDBSCAN branch: the reference's cluster_pcd (utils_cluster.py:10-29, 50-63 with
    if_hdbscan True) on crops of the demo frame pair and on a synthetic cloud with a ground mask, the
    `hdbscan` package replaced by sklearn's HDBSCAN (tools/standins/hdbscan).  Besides the labels, the SORTED
    WEIGHTS of sklearn's exact spanning tree (Prim, _single_linkage_tree_['value']) are recorded, for the
    crops and for the whole stacked demo frame (~3 min): the multiset of tree weights is the same for every
    minimum spanning tree, whatever the tie-breaking, so it pins an exact tree bit for bit."""
    import utils_cluster  # noqa: E402  (reference)
    from sklearn.cluster import HDBSCAN
    data = np.load(os.path.join(REF, "demo.npz"))
    src = data["pc1"][data["pc1_flows_valid_idx"]].astype(np.float32)
    dst = data["pc2"][data["pc2_flows_valid_idx"]].astype(np.float32)
    pts = np.concatenate([dst, src], axis=0)                                     # demo.py:210
    out = {}

    def tree_weights(p, k):
        # min_samples + 1: the hdbscan library does not count the point itself, sklearn does (tools/standins/hdbscan)
        return np.asarray(HDBSCAN(min_cluster_size=k, min_samples=k + 1, leaf_size=100).fit(p[:, :3].astype(np.float64))
                          ._single_linkage_tree_["value"], dtype=np.float64)

    crops = [(0, 5, 1500, 20, 200), (20, -10, 3000, 20, 4), (-15, 20, 2500, 30, 200)]
    for i, (cx, cy, n, k, ncl) in enumerate(crops):
        sel = (np.abs(pts[:, 0] - cx) < 15) & (np.abs(pts[:, 1] - cy) < 12)
        p = pts[sel][:n]
        a = args_ns(min_cluster_size=k, num_clusters=ncl, if_hdbscan=True, epsilon=0.25)
        lab = utils_cluster.cluster_pcd(a, p, np.ones(len(p)).astype(bool))
        out[f"crop_{i}_points"] = p
        out[f"crop_{i}_params"] = np.array([k, ncl], dtype=np.int64)
        out[f"crop_{i}_labels"] = lab.astype(np.int32)
        out[f"crop_{i}_tree_weights"] = tree_weights(p, k)
        print(f"  crop {i}: n {len(p)} k {k}: {len(np.unique(lab[lab >= 0]))} clusters kept, {int((lab == -1).sum())} unclustered")
    rng = np.random.default_rng(11)
    centers = rng.uniform(-8, 8, size=(9, 3)) * np.array([1, 1, 0.2])
    p = centers[rng.integers(0, 9, 4000)] + rng.normal(0, 0.3, size=(4000, 3)) * np.array([1, 1, 0.5])
    p[:700] = rng.uniform(-10, 10, size=(700, 3)) * np.array([1, 1, 0.2])
    p = p.astype(np.float32)
    nonground = p[:, 2] > -0.4
    a = args_ns(min_cluster_size=15, num_clusters=5, if_hdbscan=True, epsilon=0.25)
    lab = utils_cluster.cluster_pcd(a, p, nonground)
    out["synth_points"], out["synth_nonground"], out["synth_labels"] = p, nonground, lab
    out["synth_params"] = np.array([15, 5], dtype=np.int64)
    out["synth_tree_weights"] = tree_weights(p[nonground], 15)
    print(f"  synthetic: {int(nonground.sum())} non-ground of {len(p)}: {len(np.unique(lab[lab >= 0]))} clusters kept")
    import time
    t = time.time()
    m = HDBSCAN(min_cluster_size=20, min_samples=21, leaf_size=100).fit(pts.astype(np.float64))
    print(f"  demo frame: sklearn HDBSCAN of {len(pts)} points took {time.time() - t:.0f} s")
    g8 = np.load(os.path.join(OUT, "g8_demo_labels.npz"))
    lab8 = np.concatenate([g8["label_dst"], g8["label_src"]]).astype(np.int64)
    print("  labels equal the G8 label fixture:", bool(np.array_equal(m.labels_, lab8)))
    out["demo_tree_weights"] = np.asarray(m._single_linkage_tree_["value"], dtype=np.float64)
    out["demo_cpu_seconds"] = np.array(time.time() - t)
    save("g11_hdbscan", **out)


def g8_demo_mp10000():
    g8_demo(max_points=10000, name="g8_demo_mp10000")


def g8_cuda_rule():
    """The two G8 runs again with torch.topk's CUDA tie order (inputs and labels live in g8_demo.npz / g8_demo_labels.npz)."""
    g8_demo(max_points=2048, name="g8_demo_cudatopk", topk_rule="cuda")
    g8_demo(max_points=10000, name="g8_demo_mp10000_cudatopk", topk_rule="cuda")


def _ref_hist_icp_capped(a, src, dst, cap):
    """The reference's own utils_match.hist_icp + match_eval with the ICP cap of BASELINE configs 2 / 4 (<= 50
    iterations): utils_icp.pytorch3d_icp passes max_iterations=100 as a literal (utils_icp.py:54), so the imported
    `iterative_closest_point` name inside utils_icp is wrapped for the duration of the call; everything else runs
    unmodified.  -> T, init poses (estimate_init_pose on the swapped batch, what hist_icp computes), batch-global
    iteration count, converged flag, match_eval's six outputs."""
    seen = {}
    orig = utils_icp.iterative_closest_point

    def capped(*args, **kw):
        kw["max_iterations"] = cap
        r = orig(*args, **kw)
        seen["iterations"] = len(r.t_history)
        seen["converged"] = bool(r.converged)
        return r

    orig_init = utils_match.estimate_init_pose

    def spy_init(args_, s_, d_):
        r = orig_init(args_, s_, d_)
        seen["init"] = r.clone()
        return r

    utils_icp.iterative_closest_point = capped
    utils_match.estimate_init_pose = spy_init
    try:
        T = utils_match.hist_icp(a, src, dst)
    finally:
        utils_icp.iterative_closest_point = orig
        utils_match.estimate_init_pose = orig_init
    ev = utils_match.match_eval(a, src, dst, T)
    return T, seen, ev


def g12_headline():
    """G12: the reference's own Python at the sizes the bench quotes -- the whole config-2 batch
    (256 x 1024, synthetic.make_batch(256, 1024, seed=0)) and config 4's 64-pair sample (pairs 0..63 of
    make_batch(1024, 2048, seed=0)), ICP capped at 50 iterations.  Inputs are seeds (the generator is this repo's);
    stored: transforms, initial poses, the batch-global iteration count, match_eval's outputs.

    The run is repeated with 1 and 32 torch threads (torch-CPU's long fp32 sums are split by thread count, so the
    REFERENCE'S OWN result depends on it): the rows of T that differ from the 8-thread run are stored sparsely
    (`alt{n}_idx`, `alt{n}_T`, `alt{n}_iterations`) -- the scatter of the reference against itself at this size."""
    import time
    for name, B, N, nb in (("g12_config2", 256, 1024, 256), ("g12_config4_sample", 64, 2048, 1024)):
        S, D, Tt = synthetic.make_batch(nb, N, seed=0)
        src, dst = t(S[:B]), t(D[:B])
        a = args_ns(max_points=N)
        torch.set_num_threads(8)
        t0 = time.time()
        T, seen, ev = _ref_hist_icp_capped(a, src, dst, 50)
        print(f"{name}: reference hist_icp + match_eval {time.time() - t0:.0f} s, {seen['iterations']} iterations, "
              f"converged {seen['converged']}, torch threads {torch.get_num_threads()}")
        alt = {}
        for n in (1, 32):
            torch.set_num_threads(n)
            Tn, sn, _ = _ref_hist_icp_capped(a, src, dst, 50)
            assert torch.equal(sn["init"], seen["init"])
            idx = np.nonzero((Tn != T).any(dim=2).any(dim=1).numpy())[0]
            alt[f"alt{n}_idx"] = idx.astype(np.int64)
            alt[f"alt{n}_T"] = Tn.numpy()[idx]
            alt[f"alt{n}_iterations"] = np.array(sn["iterations"])
            print(f"  {n} threads: {sn['iterations']} iterations, {len(idx)} of {B} transforms differ in some bit from the 8-thread run")
        torch.set_num_threads(8)
        save(name, seed=np.array(0), num_pairs=np.array(B), max_points=np.array(N), make_batch_pairs=np.array(nb),
             icp_max_iterations=np.array(50), torch_threads=np.array(8),
             T_hist_icp=T.numpy(), T_init=seen["init"].numpy(), icp_iterations=np.array(seen["iterations"]),
             icp_converged=np.array(seen["converged"]), cut_tied=cut_is_tied(a, src, dst, True),
             ev_errors=ev[0].numpy(), ev_inliers=ev[1].numpy(), ev_ratios=ev[2].numpy(),
             ev_ious=ev[3].numpy(), ev_translations=ev[4].numpy(), ev_rotations=ev[5].numpy(), **alt)


GENS = dict(g12=g12_headline, g8cuda=g8_cuda_rule, g8mp10000=g8_demo_mp10000, g11=g11_hdbscan, g10=g10_dbscan, g9=g9_epe, g1=g1_hist, g3=g3_nn, g4=g4_init_pose, g5=g5_icp, g6=g6_hist_icp, g8=g8_demo)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ns = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    want = [s for s in ns.only.split(",") if s] or [k for k in GENS if k not in ("g8", "g8mp10000", "g8cuda", "g9", "g10", "g11", "g12")]   # g8: ~3 min, on request
    for k in want:
        print(f"== {k}")
        GENS[k]()
