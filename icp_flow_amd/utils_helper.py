"""Drop-ins for the hot-path helpers of the reference's utils_helper.py."""
import torch

from . import _lib

PAD_VALUE = 1e8      # utils_helper.py:192


def nearest_neighbor_batch(src, dst):
    """utils_helper.py:20-30: brute-force K=1 NN over ALL rows of both clouds (pads
    included, exactly like the reference's un-lengthed knn_points call).
    src [B,N,>=3], dst [B,M,>=3] float32 -> (idx int64 [B,N], Euclidean dist float32 [B,N])."""
    assert src.dim() == 3
    assert dst.dim() == 3
    assert len(src) == len(dst)
    assert src.shape[2] >= 3
    assert dst.shape[2] >= 3
    _lib.require_gpu(src, dst)
    if src.dtype != torch.float32 or dst.dtype != torch.float32:
        raise RuntimeError("nearest_neighbor_batch: float32 only")
    # the reference slices [:, :, 0:3]; here any row stride >= 3 is scanned in place
    q = src if src.is_contiguous() else src.contiguous()
    t = dst if dst.is_contiguous() else dst.contiguous()
    B, NQ, sq = q.shape
    _, NT, st = t.shape
    idx = torch.empty((B, NQ), dtype=torch.int64, device=q.device)
    dist = torch.empty((B, NQ), dtype=torch.float32, device=q.device)
    _lib.call("icpflow_nn_batch", _lib.ptr(q), _lib.ptr(t), B, NQ, NT, sq, st, None, None, 1,
              _lib.ptr(idx), _lib.ptr(dist), _lib.stream(q.device))
    return idx, dist


def knn_points_lengths(p1, p2, lengths1, lengths2):
    """pytorch3d.ops.knn_points(p1, p2, lengths1, lengths2, K=1) as called by the ICP loop
    (utils_icp_pytorch3d.py:154-156): -> (squared dists [B,N], idx int64 [B,N])."""
    _lib.require_gpu(p1, p2, lengths1, lengths2)
    q, t = p1.contiguous(), p2.contiguous()
    B, NQ, sq = q.shape
    _, NT, st = t.shape
    l1 = lengths1.to(torch.int32).contiguous()
    l2 = lengths2.to(torch.int32).contiguous()
    idx = torch.empty((B, NQ), dtype=torch.int64, device=q.device)
    d2 = torch.empty((B, NQ), dtype=torch.float32, device=q.device)
    _lib.call("icpflow_nn_batch", _lib.ptr(q), _lib.ptr(t), B, NQ, NT, sq, st, _lib.ptr(l1), _lib.ptr(l2),
              0, _lib.ptr(idx), _lib.ptr(d2), _lib.stream(q.device))
    return d2, idx


def transform_points_batch(xyz, pose):
    """utils_helper.py:76-87: [x y z 1] @ pose^T with the flag column carried through."""
    assert xyz.dim() == 3
    assert pose.dim() == 3
    assert xyz.shape[2] == 4
    assert pose.shape[1] == 4
    assert pose.shape[2] == 4
    assert len(xyz) == len(pose)
    x = _lib.cloud(xyz, "xyz")
    P = pose.to(torch.float32).contiguous()
    _lib.require_gpu(P)
    out = torch.empty_like(x)
    _lib.call("icpflow_transform_points", _lib.ptr(x), _lib.ptr(P), x.shape[0], x.shape[1],
              _lib.ptr(out), _lib.stream(x.device))
    return out


def count_valid(cloud):
    """Valid points per pair, int32 [B] (mask.sum(dim=1) of the reference, on device)."""
    x = _lib.cloud(cloud)
    n = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
    _lib.call("icpflow_count_valid", _lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(n), _lib.stream(x.device))
    return n


def random_choice(m, n):
    """utils_helper.py:198-201 (same torch RNG stream as the reference)."""
    assert m >= n
    return torch.randperm(m)[0:n]


def pad_segment(seg, max_points):
    """utils_helper.py:185-196: [n,3] -> [max_points,4]; pads are (1e8,1e8,1e8,0), an
    over-long segment is randomly subsampled.  Pure layout work, plain torch ops."""
    n = len(seg)
    flag = seg.new_ones((max_points, 1))
    if n > max_points:
        seg = seg[random_choice(n, max_points).to(seg.device), :]
    elif n < max_points:
        flag[n:] = 0.0
        seg = torch.cat([seg, seg.new_full((max_points - n, 3), PAD_VALUE)], dim=0)
    assert len(seg) == max_points
    return torch.cat([seg, flag], dim=1)
