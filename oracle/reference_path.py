"""CPU restatement of ICP-Flow's cluster-pair registration hot path.

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Every function names the reference lines it follows (paths are relative to
/root/reference).  The arithmetic is torch-CPU float32 in the reference's
operation order; the two device primitives (vote, K=1 brute-force NN) live in
oracle_core.c.  The restatement is pinned by tests/golden/*.npz, which were
produced by importing the reference's OWN vendored Python on CPU tensors
(tools/gen_golden.py) -- see tests/test_oracle_golden.py.

Third-party arithmetic that is not under /root/reference and is therefore
restated from its published behaviour ("parity unpinned" at that boundary):
pytorch3d 0.7.4 knn_points / wmean / matrix_to_euler_angles.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import core

PAD_VALUE = 1e8          # utils_helper.py:192
ICP_MAX_ITER = 100       # utils_icp.py:54
ICP_REL_RMSE = 1e-6      # utils_icp.py:55
TOPK = 5                 # utils_hist.py:21
NMS_KERNEL = 11          # utils_hist.py:21


def default_args(**kw):
    """Namespace with the fields the hot path reads (demo.sh:9-13 values)."""
    a = dict(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=1024,
             thres_iou=0.2, thres_rot=0.1, thres_error=0.2, thres_box=0.1,
             min_cluster_size=20)
    a.update(kw)
    return SimpleNamespace(**a)


# --------------------------------------------------------------------------
# input format, utils_helper.py:185-201
# --------------------------------------------------------------------------
def pad_segment(seg, max_points):
    """[n,3] -> [max_points,4]; pads are (1e8,1e8,1e8,0); n>max: randperm subsample."""
    n = len(seg)
    flag = seg.new_ones((max_points, 1))
    if n > max_points:
        keep = torch.randperm(n)[0:max_points]              # utils_helper.py:198-201
        seg = seg[keep, :]
    elif n < max_points:
        flag[n:] = 0.0
        seg = torch.cat([seg, seg.new_full((max_points - n, 3), PAD_VALUE)], dim=0)
    return torch.cat([seg, flag], dim=1)


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def hist(X, Y, min_x, min_y, min_z, max_x, max_y, max_z, len_x, len_y, len_z, mini_batch=8):
    """hist_cuda/hist.py:39-51 -> hist_cuda_core.cuh:40-60 (oracle_core.c)."""
    return core.hist_vote(X, Y, (min_x, min_y, min_z), (max_x, max_y, max_z),
                          (len_x, len_y, len_z))


def knn_points(p1, p2, lengths1=None, lengths2=None, return_nn=False):
    """pytorch3d.ops.knn_points(K=1): (dists [B,N1], idx [B,N1], nn [B,N1,3]|None)."""
    return core.knn1(p1, p2, lengths1, lengths2, return_nn)


def nearest_neighbor_batch(src, dst):
    """utils_helper.py:20-30: un-lengthed K=1 NN, Euclidean (sqrt) distance."""
    assert src.dim() == 3 and dst.dim() == 3 and len(src) == len(dst)
    d2, idx, _ = knn_points(src[:, :, 0:3], dst[:, :, 0:3])
    return idx, d2.sqrt()


def _fma32(a, b, c):
    """fp32 fused multiply-add: the product of two fp32 numbers is exact in fp64, the sum is rounded to fp64 and then
    to fp32 (the double rounding can only matter on an exact fp32 tie of a 53-bit sum)."""
    return (a.double() * b.double() + c.double()).to(a.dtype)


def point_mm(P, M):
    """`torch.bmm(P, M)` for rows of points P [B,N,K] times a small matrix M [B,K,K'] (K = 3 or 4), with the rounding
    PINNED: out = fma(p_{K-1}, m_{K-1}, ... fma(p_1, m_1, p_0 * m_0)), k ascending.
    A BLAS call does not define its rounding: torch-CPU runs this very product as that FMA chain on the AVX2 host the
    golden fixtures were generated on (and for the padded sizes they use, N >= 48), and as separate multiplies and adds
    on other hosts (measured on the Zen-5 host of the GPU boxes) or for tiny N -- 1 ulp of the moved point apart on
    ~40 % of the rows, enough to hand a query with two nearly equidistant targets the other neighbour.  The restatement
    must not depend on the machine it runs on: it takes the fixtures' arithmetic, which is also the kernels'
    (icp.hip: fmaf chain; nvcc contracts the same way)."""
    if P.dtype != torch.float32:
        return torch.bmm(P, M)
    acc = P[..., :, 0:1] * M[..., 0:1, :]
    for k in range(1, P.shape[-1]):
        acc = _fma32(P[..., :, k:k + 1], M[..., k:k + 1, :], acc)
    return acc


def tiny_mm(A, M):
    """`torch.bmm` of two SMALL matrices ([.,3,3], [.,4,4], a [.,1,3] row): products and sums rounded one by one, k
    ascending -- how torch-CPU evaluates them on the host the fixtures come from (its native small-matrix path).
    Pinned for the same reason as point_mm."""
    if A.dtype != torch.float32:
        return torch.bmm(A, M)
    acc = A[..., :, 0:1] * M[..., 0:1, :]
    for k in range(1, A.shape[-1]):
        acc = acc + A[..., :, k:k + 1] * M[..., k:k + 1, :]
    return acc


def transform_points_batch(xyz, pose):
    """utils_helper.py:76-87: [x y z 1] @ pose^T, flag column carried through."""
    b, n, _ = xyz.shape
    hom = torch.cat([xyz[:, :, 0:3], xyz.new_ones((b, n, 1))], dim=-1)
    moved = point_mm(hom, pose.transpose(1, 2))
    return torch.cat([moved[:, :, 0:3], xyz[:, :, 3:4]], dim=-1)


def tree_sum(x, dim):
    """Sum over `dim` in balanced pairwise order (halves added elementwise, zero padded to a power of two), in
    x's dtype.  NOT torch's order: torch-CPU accumulates long runs sequentially per output element (error grows
    like sqrt(n) ulps of the running total), GPU reductions -- the reference runs on CUDA -- are trees (log n).
    Tests use this to show which of the reference's observables depend on the summation ORDER of its backend."""
    x = x.movedim(dim, 0)
    n = x.shape[0]
    m = 1
    while m < n:
        m <<= 1
    if m != n:
        x = torch.cat([x, x.new_zeros((m - n,) + tuple(x.shape[1:]))], dim=0)
    while x.shape[0] > 1:
        h = x.shape[0] // 2
        x = x[:h] + x[h:]
    return x[0]


def wmean(x, weight, eps=1e-9, sum_order=None):
    """pytorch3d.ops.utils.wmean(dim=-2, keepdim=True); weight may be bool."""
    w = weight[..., None]
    if sum_order == "tree":
        return tree_sum(x * w, -2).unsqueeze(-2) / w.sum(dim=-2, keepdim=True).clamp(eps)
    return (x * w).sum(dim=-2, keepdim=True) / w.sum(dim=-2, keepdim=True).clamp(eps)


def matrix_to_euler_zyx(M):
    """pytorch3d.transforms.matrix_to_euler_angles(M, 'ZYX') in radians."""
    return torch.stack([torch.atan2(M[..., 1, 0], M[..., 0, 0]),
                        torch.asin(-M[..., 2, 0]),
                        torch.atan2(M[..., 2, 1], M[..., 2, 2])], dim=-1)


# --------------------------------------------------------------------------
# translation histogram -> initial pose, utils_hist.py
# --------------------------------------------------------------------------
def bin_edges(args):
    """utils_hist.py:61-65 (float32 torch.arange)."""
    tf, th, eps = args.translation_frame, args.thres_dist, 1e-8
    ex = torch.arange(-tf, tf + th - eps, th)
    ey = torch.arange(-tf, tf + th - eps, th)
    ez = torch.arange(-th, th + th - eps, th)
    return ex, ey, ez


def nms_mask(x, kernel_size=NMS_KERNEL):
    """utils_hist.py:22-26: votes surviving the 3-D max-pool test, else 0."""
    xp = torch.nn.functional.max_pool3d(x[:, None], kernel_size=kernel_size, stride=1,
                                        padding=(kernel_size - 1) // 2)
    return (x[:, None] * (x[:, None] == xp).float())[:, 0]


def topk_nms(x, k=TOPK, kernel_size=NMS_KERNEL):
    """utils_hist.py:21-29.  torch.topk's order among equal votes is
    implementation-defined; the build's deterministic rule is
    (vote descending, flat index ascending) -- a stable sort gives exactly that."""
    b = x.shape[0]
    flat = nms_mask(x, kernel_size).reshape(b, -1)
    order = torch.argsort(flat, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(flat, 1, order), order.long()


def decode_peaks(args, idx, shape, edges):
    """utils_hist.py:78: flat index -> left bin edge (+ thres_dist//2 == 0.0)."""
    _, h, w, d = shape
    ex, ey, ez = edges
    return torch.stack([ex[idx // d // w % h], ey[idx // d % w], ez[idx % d]], dim=-1) \
        + args.thres_dist // 2


def estimate_init_pose_batch(args, src, dst, return_aux=False):
    """utils_hist.py:46-124."""
    p1, p2 = src[:, :, 0:3], dst[:, :, 0:3]
    m1, m2 = src[:, :, -1] > 0.0, dst[:, :, -1] > 0.0
    ex, ey, ez = bin_edges(args)
    votes = hist(dst, src, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(),
                 len(ex), len(ey), len(ez))                                  # :69-72
    b = votes.shape[0]
    peak_votes, peak_idx = topk_nms(votes)                                    # :77
    t_peaks = decode_peaks(args, peak_idx, votes.shape, (ex, ey, ez))         # :78
    n = p1.shape[1]
    cand = torch.cat([t_peaks, t_peaks.new_zeros(b, 1, 3)], dim=1)            # :83 (zero LAST)
    k = cand.shape[1]
    moved = (p1[:, None] + cand[:, :, None, :]).reshape(b * k, n, 3)          # :86
    fixed = p2[:, None].expand(-1, k, -1, -1).reshape(b * k, n, 3)            # :87
    _, e_fwd = nearest_neighbor_batch(moved, fixed)                           # :89
    _, e_bwd = nearest_neighbor_batch(fixed, moved)                           # :90
    e_fwd = (e_fwd.view(b, k, n) * m1[:, None]).sum(-1) / m1[:, None].sum(-1)   # :101
    e_bwd = (e_bwd.view(b, k, n) * m2[:, None]).sum(-1) / m2[:, None].sum(-1)   # :102
    score = torch.minimum(e_fwd, e_bwd)                                       # :103
    _, pick = score.min(dim=-1)                                               # :104
    t_best = cand[torch.arange(b), pick]                                      # :106
    T = torch.eye(4)[None].repeat(b, 1, 1)                                    # :121
    T[:, 0:3, 3] = t_best
    if return_aux:
        return T, dict(votes=votes, peak_votes=peak_votes, peak_idx=peak_idx,
                       candidates=cand, score=score, pick=pick)
    return T


def estimate_init_pose(args, src, dst):
    """utils_hist.py:33-44: chunk_size pairs at a time (results chunk-independent)."""
    assert len(src) == len(dst)
    out = [estimate_init_pose_batch(args, src[s:s + args.chunk_size], dst[s:s + args.chunk_size])
           for s in range(0, len(src), args.chunk_size)]
    return torch.vstack(out)


# --------------------------------------------------------------------------
# ICP, utils_icp_pytorch3d.py
# --------------------------------------------------------------------------
def corresponding_points_alignment(X, Y, weights, eps=1e-9, dtype=None, sum_order=None, allow_reflection=False,
                                   estimate_scale=False):
    """utils_icp_pytorch3d.py:303-382 (ICP-Flow: estimate_scale=False, allow_reflection=False; with estimate_scale
    the return value is (R, T, s)).
    X, Y [B,N,3] already mask-multiplied, weights bool [B,N].  y = x R + T.

    dtype=torch.float64 is NOT the reference: it evaluates the same formulas on the same
    fp32 inputs in double (centroids, covariance, SVD) and rounds R, T to fp32 at the end.
    Tests use it to separate "the kernel implements this algorithm" (tight tolerance against
    the fp64 evaluation) from "how far the reference's own fp32 reductions are from it"."""
    out_dtype = X.dtype
    if dtype is not None:
        X, Y = X.to(dtype), Y.to(dtype)
    b = X.shape[0]
    mu_x = wmean(X, weights, eps, sum_order)                                  # :314
    mu_y = wmean(Y, weights, eps, sum_order)                                  # :315
    w = weights[:, :, None]
    Xc = (X - mu_x) * w                                                       # :318,324
    Yc = (Y - mu_y) * w                                                       # :319,325
    total = torch.clamp(weights.sum(1), eps)                                  # :326
    if sum_order == "tree":   # the same 3x3 product, its sum over the points in pairwise order (see tree_sum)
        H = tree_sum(Xc[:, :, :, None] * Yc[:, :, None, :], 1) / total[:, None, None]
    else:
        H = torch.bmm(Xc.transpose(2, 1), Yc) / total[:, None, None]          # :335-336
    U, S, V = torch.svd(H)                                                    # :339
    E = torch.eye(3, dtype=H.dtype)[None].repeat(b, 1, 1)
    if not allow_reflection:                                                  # :354
        E[:, -1, -1] = torch.det(tiny_mm(U, V.transpose(2, 1)))               # :358-359
    R = tiny_mm(tiny_mm(U, E), V.transpose(2, 1))                             # :362
    if estimate_scale:
        trace_ES = (torch.diagonal(E, dim1=1, dim2=2) * S).sum(1)             # :366
        Xcov = (Xc * Xc).sum((1, 2)) / total                                  # :367
        s = trace_ES / torch.clamp(Xcov, eps)                                 # :370
        T = mu_y[:, 0, :] - s[:, None] * tiny_mm(mu_x, R)[:, 0, :]            # :373
        return R.to(out_dtype), T.to(out_dtype), s.to(out_dtype)
    T = mu_y[:, 0, :] - tiny_mm(mu_x, R)[:, 0, :]                             # :376
    return R.to(out_dtype), T.to(out_dtype)


def iterative_closest_point(X, Y, thres=0.1, max_iterations=ICP_MAX_ITER,
                            relative_rmse_thr=ICP_REL_RMSE, trace=False, kabsch_dtype=None, sum_order=None,
                            init_transform=None, allow_reflection=False, estimate_scale=False):
    """utils_icp_pytorch3d.py:100-225.  Returns a namespace with
    converged, rmse, Xt, R, T, iterations (number of loop bodies executed) and,
    with trace=True, the per-iteration (R, T, rmse, inlier count) history.
    kabsch_dtype: see corresponding_points_alignment (None = the reference's fp32); sum_order: see tree_sum
    (None = torch's own order)."""
    X0 = X[:, :, 0:3].clone()                                                 # :100,115
    Yt = Y[:, :, 0:3]
    b = X0.shape[0]
    m0 = X[:, :, -1] > 0.0                                                    # :109,116
    n_x = m0.sum(-1)                                                          # :111
    n_y = (Y[:, :, -1] > 0.0).sum(-1)                                         # :112
    Xt = X0
    R = torch.eye(3)[None].repeat(b, 1, 1)                                    # :140
    T = X0.new_zeros((b, 3))
    s = X0.new_ones(b)
    if init_transform is not None:                                            # :118-138
        R, T = init_transform[0], init_transform[1]
        if len(init_transform) > 2:
            s = init_transform[2]
        Xt = s[:, None, None] * point_mm(X0, R) + T[:, None, :]               # _apply_similarity_transform, :395
    prev = None
    rmse = None
    converged = False
    it = -1
    history = []
    thr2 = thres ** 2                       # python double; torch compares in fp32 (:160)
    for it in range(max_iterations):                                          # :153
        d2, _, nn = knn_points(Xt, Yt, n_x, n_y, return_nn=True)              # :154-157
        w = torch.logical_and(m0, d2 <= thr2)                                 # :160-161
        sol = corresponding_points_alignment(X0 * w[:, :, None], nn * w[:, :, None], w,
                                             dtype=kabsch_dtype, sum_order=sum_order,
                                             allow_reflection=allow_reflection, estimate_scale=estimate_scale)
        R, T = sol[0], sol[1]
        s = sol[2] if estimate_scale else X0.new_ones(b)                      # (:376-379: unit scale when not estimated)
        Xt = s[:, None, None] * point_mm(X0, R) + T[:, None, :]               # :177,395
        sq = ((Xt - nn) ** 2).sum(2)                                          # :191
        if kabsch_dtype is not None:
            sq = sq.to(kabsch_dtype)
        rmse = wmean(sq[:, :, None], w, sum_order=sum_order).sqrt()[:, 0, 0].to(X0.dtype)   # :192
        rel = rmse.new_ones(b) if prev is None else (prev - rmse) / prev      # :195-198
        if trace:
            history.append((R.clone(), T.clone(), rmse.clone(), w.sum(-1).clone()))
        if bool((rel <= relative_rmse_thr).all()):                            # :209
            converged = True
            break
        prev = rmse
    return SimpleNamespace(converged=converged, rmse=rmse, Xt=Xt, R=R, T=T, s=s,
                           iterations=it + 1, history=history)


def pytorch3d_icp(args, src, dst, max_iterations=ICP_MAX_ITER, kabsch_dtype=None, sum_order=None):
    """utils_icp.py:50-73: column-vector 4x4 from the row-vector (R, T)."""
    sol = iterative_closest_point(src, dst, thres=args.thres_dist,
                                  max_iterations=max_iterations,
                                  relative_rmse_thr=ICP_REL_RMSE, kabsch_dtype=kabsch_dtype, sum_order=sum_order)
    b = len(sol.T)
    M = torch.zeros(b, 4, 4)
    M[:, 0:3, 0:3] = sol.R.transpose(1, 2)                                    # :63-64
    M[:, 0:3, 3] = sol.T
    M[:, 3, 3] = 1.0                                                          # :65
    return M, sol


def apply_icp(args, src, dst, init_poses, max_iterations=ICP_MAX_ITER, return_aux=False, kabsch_dtype=None,
              sum_order=None):
    """utils_icp.py:20-48: ICP from the init pose, roll back where it did not help.
    kabsch_dtype: see corresponding_points_alignment (None = the reference's fp32)."""
    moved = transform_points_batch(src, init_poses)                           # :21
    M, sol = pytorch3d_icp(args, moved, dst, max_iterations, kabsch_dtype, sum_order)   # :23
    M = tiny_mm(M, init_poses)                                                # :24
    valid = src[:, :, -1] > 0.0                                               # :27
    _, e0 = nearest_neighbor_batch(moved, dst)                                # :28
    e0 = (e0 * valid).sum(1) / valid.sum(1)                                   # :29
    _, e1 = nearest_neighbor_batch(transform_points_batch(src, M), dst)       # :31-32
    e1 = (e1 * valid).sum(1) / valid.sum(1)                                   # :33
    worse = e1 >= e0                                                          # :34
    M[worse] = init_poses[worse]                                              # :35
    if return_aux:
        return M, dict(error_init=e0, error_icp=e1, rolled_back=worse,
                       iterations=sol.iterations, converged=sol.converged)
    return M


# --------------------------------------------------------------------------
# orchestration + metrics, utils_match.py
# --------------------------------------------------------------------------
def hist_icp(args, src, dst, max_iterations=ICP_MAX_ITER, return_aux=False, kabsch_dtype=None, init=None,
             sum_order=None):
    """utils_match.py:138-157.  kabsch_dtype: see corresponding_points_alignment (None = the reference's fp32);
    init: initial poses of a previous call on the same batch (skips the vote and the scoring scans)."""
    n1 = (src[:, :, -1] > 0.0).sum(1)
    n2 = (dst[:, :, -1] > 0.0).sum(1)
    swap = n1 > n2                                                            # :142 (strict)
    a, b = src.clone(), dst.clone()
    a[swap] = dst[swap]
    b[swap] = src[swap]
    if init is None:
        init = estimate_init_pose(args, a, b)                                 # :149
    if return_aux:
        M, aux = apply_icp(args, a, b, init, max_iterations, return_aux=True, kabsch_dtype=kabsch_dtype,
                           sum_order=sum_order)
    else:
        M, aux = apply_icp(args, a, b, init, max_iterations, kabsch_dtype=kabsch_dtype, sum_order=sum_order), None   # :150
    if int(swap.sum()) > 0:                                                   # :152
        M = M.clone()
        M[swap] = torch.linalg.inv(M[swap])                                   # :154
    if return_aux:
        aux.update(init=init, swapped=swap)
        return M, aux
    return M


def match_eval(args, pcd1, pcd2, transformations):
    """utils_match.py:159-213 -> errors, inliers, ratios, ious [B,2]; translations, rotations [B,3]."""
    moved = transform_points_batch(pcd1, transformations)                     # :160
    m1 = pcd1[:, :, -1] > 0.0
    m2 = pcd2[:, :, -1] > 0.0
    _, e1 = nearest_neighbor_batch(moved, pcd2)                               # :165
    _, e2 = nearest_neighbor_batch(pcd2, moved)                               # :166
    in1 = torch.logical_and(e1 < args.thres_dist, m1).float()                 # :168 (strict <)
    in2 = torch.logical_and(e2 < args.thres_dist, m2).float()                 # :169
    c1, c2 = m1.sum(1), m2.sum(1)
    ratio1 = in1.sum(1) / c1                                                  # :171
    ratio2 = in2.sum(1) / c2                                                  # :172
    iou1 = in1.sum(1) / (c1 + c2 - in2.sum(1))                                # :174
    iou2 = in2.sum(1) / (c1 + c2 - in1.sum(1))                                # :175
    err1 = (e1 * m1).sum(1) / c1                                              # :177
    err2 = (e2 * m2).sum(1) / c2                                              # :178
    mean_moved = (moved[:, :, 0:3] * m1[:, :, None]).sum(1) / m1.sum(1, keepdim=True)   # :180
    mean_orig = (pcd1[:, :, 0:3] * m1[:, :, None]).sum(1) / m1.sum(1, keepdim=True)     # :181
    translations = mean_moved - mean_orig                                     # :183
    rotations = matrix_to_euler_zyx(transformations[:, 0:3, 0:3]) * 180.0 / np.pi      # :184
    return (torch.stack([err1, err2], 1), torch.stack([in1.sum(1), in2.sum(1)], 1),
            torch.stack([ratio1, ratio2], 1), torch.stack([iou1, iou2], 1),
            translations, rotations)


# --------------------------------------------------------------------------
# the caller of the path: cluster association and per-point flow
# utils_check.py, utils_match.py:24-136, utils_helper.py:108-115,166-183, utils_flow.py:57-69
# --------------------------------------------------------------------------
def get_bbox_tensor(points):
    """utils_helper.py:166-170: sorted axis-aligned extents."""
    ext = [torch.abs(points[:, k].max() - points[:, k].min()) for k in range(3)]
    return sorted(ext)


def sanity_check(args, src_points, dst_points, src_labels, dst_labels, pairs):
    """utils_check.py:21-49."""
    keep = []
    for pair in pairs:
        src = src_points[src_labels == pair[0]]
        dst = dst_points[dst_labels == pair[1]]
        if min(len(src), len(dst)) < args.min_cluster_size:                       # :31
            continue
        if min(pair[0], pair[1]) < 0:                                             # :32
            continue
        if torch.linalg.norm((dst.mean(0) - src.mean(0))[0:2]) > args.translation_frame:   # :36
            continue
        bs, bd = get_bbox_tensor(src), get_bbox_tensor(dst)
        if any(min(bs[k], bd[k]) < args.thres_box * max(bs[k], bd[k]) for k in range(3)):  # :41-43
            continue
        keep.append(pair)
    return torch.vstack(keep) if keep else torch.zeros((0, 2))


def check_transformation(args, translation, rotation, iou):
    """utils_check.py:51-66."""
    if torch.linalg.norm(translation) > args.translation_frame:
        return False
    if iou < args.thres_iou:
        return False
    if torch.abs(rotation[1:3]).max() > args.thres_rot * 90.0:
        return False
    return True


def match_pairs(args, src_points, dst_points, src_labels, dst_labels, pairs):
    """utils_match.py:69-136."""
    su, du = torch.unique(src_labels), torch.unique(dst_labels)
    m_err = torch.zeros((len(su), len(du), 2)) + 1e8
    m_inl = torch.zeros((len(su), len(du), 2))
    m_rat = torch.zeros((len(su), len(du), 2))
    m_iou = torch.zeros((len(su), len(du), 2))
    m_T = torch.zeros((len(su), len(du), 4, 4))
    assert len(pairs) > 0
    segs_src, segs_dst = [], []
    for pair in pairs:                                                            # :81-91
        segs_src.append(pad_segment(src_points[src_labels == pair[0], 0:3], args.max_points))
        segs_dst.append(pad_segment(dst_points[dst_labels == pair[1], 0:3], args.max_points))
    segs_src, segs_dst = torch.stack(segs_src), torch.stack(segs_dst)
    T = hist_icp(args, segs_src, segs_dst)                                        # :92
    errors, inliers, ratios, ious, translations, rotations = match_eval(args, segs_src, segs_dst, T)   # :93
    matches = 0
    for k, pair in enumerate(pairs):                                              # :96-108
        if not check_transformation(args, translations[k], rotations[k], min(ious[k])):
            continue
        i = torch.nonzero(su == pair[0])
        j = torch.nonzero(du == pair[1])
        m_err[i, j, :] = errors[k]
        m_inl[i, j, :] = inliers[k]
        m_rat[i, j, :] = ratios[k]
        m_iou[i, j, :] = ious[k]
        m_T[i, j] = T[k]
        matches += 1
    if matches == 0:
        return torch.zeros((0, 10)), torch.zeros((0, 4, 4))
    err_min = m_err.min(-1)[0]
    rows = torch.arange(0, len(err_min))                                          # utils_helper.py:108-110
    cols = torch.argmin(err_min, dim=1)
    ok = err_min[rows, cols] < args.thres_error                                   # :112
    rows, cols = rows[ok], cols[ok]
    out = torch.cat([su[rows][:, None], du[cols][:, None], m_err[rows, cols], m_inl[rows, cols],
                     m_rat[rows, cols], m_iou[rows, cols]], dim=1)                # :123-128
    return out, m_T[rows, cols]


def setdiff1d(t1, t2):
    """utils_helper.py:172-183."""
    t12, counts = torch.cat([torch.unique(t1), torch.unique(t2)]).unique(return_counts=True)
    return t12[torch.where(counts.eq(1))]


def match_pcds(args, src_points, dst_points, src_labels, dst_labels):
    """utils_match.py:24-66."""
    su = torch.unique(src_labels).long()
    du = torch.unique(dst_labels).long()
    lu = torch.unique(torch.cat([su, du]))
    pairs = torch.stack([lu, lu], dim=1)
    pairs = pairs[pairs.min(dim=1)[0] >= 0]                                       # :30-31
    true = sanity_check(args, src_points, dst_points, src_labels, dst_labels, pairs)
    if len(true) > 0:
        p_sta, T_sta = match_pairs(args, src_points, dst_points, src_labels, dst_labels, true)
    else:
        p_sta, T_sta = torch.zeros((0, 10)), torch.zeros((0, 4, 4))
    if len(p_sta) < len(lu):                                                      # :45
        if len(p_sta) > 0:
            su = setdiff1d(su, p_sta[:, 0])
            du = setdiff1d(du, p_sta[:, 1])
        pairs = torch.stack([su.repeat_interleave(len(du)), du.repeat(len(su))], dim=1)
        true = sanity_check(args, src_points, dst_points, src_labels, dst_labels, pairs)
    else:
        true = torch.zeros(0, 2)
    if len(true) > 0:
        p_dyn, T_dyn = match_pairs(args, src_points, dst_points, src_labels, dst_labels, true)
    else:
        p_dyn, T_dyn = torch.zeros((0, 10)), torch.zeros((0, 4, 4))
    return torch.cat([p_sta, p_dyn], dim=0), torch.cat([T_sta, T_dyn], dim=0)


def flow_estimation_torch(src_points, src_labels, pairs, transformations, pose):
    """utils_flow.py:57-69."""
    n = len(src_points)
    T = torch.eye(4)[None].repeat(n, 1, 1)
    rows, cols = torch.nonzero((src_labels[:, None] - pairs[:, 0][None, :]) == 0, as_tuple=True)
    T[rows] = transformations[cols]
    T = tiny_mm(T, pose[None].expand(n, 4, 4))
    hom = torch.cat([src_points, src_points.new_ones(n, 1)], dim=-1)
    return tiny_mm(T, hom[:, :, None])[:, 0:3, 0] - src_points
