"""Drop-ins for the registration entry points of the reference's utils_match.py."""
import torch

from . import _lib
from .utils_check import ClusterTable, check_transformation, sanity_check
from .utils_hist import bin_edges
from .utils_icp import _icp_options


def hist_icp(args, src, dst, return_iterations=False):
    """utils_match.py:138-157 -- one full registration per cluster pair in ONE call:
    smaller-cloud-first swap, histogram initial pose, ICP with roll-back, inverse for the
    swapped pairs.  src, dst: float32 [B, max_points, 4] -> transforms float32 [B,4,4]."""
    s = _lib.cloud(src, "src")
    d = _lib.cloud(dst, "dst")
    assert s.shape == d.shape, "src and dst must share [B, max_points, 4]"
    B, N, _ = s.shape
    ex, ey, ez = bin_edges(args, s.device)
    lens = (len(ex), len(ey), len(ez))
    max_it, rel, stop = _icp_options(args)
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=s.device)
    iters = torch.zeros((1,), dtype=torch.int32, device=s.device)
    ws = _lib.workspace(s.device, _lib.workspace_bytes(B, N, lens))
    _lib.call("icpflow_hist_icp", _lib.ptr(s), _lib.ptr(d), B, N, _lib.ptr(ex), lens[0], _lib.ptr(ey),
              lens[1], _lib.ptr(ez), lens[2], float(args.thres_dist // 2), float(args.thres_dist), max_it,
              rel, stop, _lib.ptr(out), _lib.ptr(iters), _lib.ptr(ws), ws.numel(), _lib.stream(s.device))
    return (out, iters) if return_iterations else out


def match_eval(args, pcd1, pcd2, transformations):
    """utils_match.py:159-213 -> (errors, inliers, ratios, ious) [B,2], translations [B,3],
    rotations [B,3] (Euler ZYX degrees)."""
    a = _lib.cloud(pcd1, "pcd1")
    b = _lib.cloud(pcd2, "pcd2")
    assert a.shape == b.shape
    B, N, _ = a.shape
    T = transformations.to(device=a.device, dtype=torch.float32).contiguous()
    assert T.shape == (B, 4, 4)
    dev = a.device
    o2 = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(4)]
    o3 = [torch.empty((B, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    ws = _lib.workspace(dev, _lib.workspace_bytes(B, N))
    _lib.call("icpflow_match_eval", _lib.ptr(a), _lib.ptr(b), _lib.ptr(T), B, N, float(args.thres_dist),
              _lib.ptr(o2[0]), _lib.ptr(o2[1]), _lib.ptr(o2[2]), _lib.ptr(o2[3]), _lib.ptr(o3[0]),
              _lib.ptr(o3[1]), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
    return o2[0], o2[1], o2[2], o2[3], o3[0], o3[1]


# --------------------------------------------------------------------------------------
# the caller of the registration path: cluster association
# --------------------------------------------------------------------------------------
def _padded_batch(args, table, labels_wanted):
    """pad_segment (utils_helper.py:185-196) of every requested cluster, as one gather:
    -> [B, max_points, 4] on the device.  Over-long clusters are subsampled with
    torch.randperm on the host generator, one draw per cluster in request order -- the same
    stream of draws the reference makes (utils_helper.py:198-201)."""
    N = int(args.max_points)
    idx = table.find(labels_wanted)
    assert bool((idx >= 0).all())
    start = table.start[idx]
    count = table.count[idx]
    dev = table.points.device
    slot = torch.arange(N, device=dev)[None, :]
    rows = torch.where(slot < count[:, None], start[:, None] + slot, torch.full_like(slot, -1))
    return rows, count


def match_pairs(args, src_points, dst_points, src_labels, dst_labels, pairs, tables=None):
    """utils_match.py:69-136: register every candidate pair, reject implausible transforms
    (check_transformation), assign each source cluster its best destination cluster (row arg-min
    of min(err_src, err_dst) below thres_error).  -> pairs [P,10], transformations [P,4,4]."""
    assert len(pairs) > 0
    dev = src_points.device
    st, dt = tables if tables is not None else (ClusterTable(src_points, src_labels), ClusterTable(dst_points, dst_labels))
    N = int(args.max_points)
    rows_s, cnt_s = _padded_batch(args, st, pairs[:, 0])
    rows_d, cnt_d = _padded_batch(args, dt, pairs[:, 1])
    # random subsample of over-long clusters: src then dst, pair by pair (reference order)
    over_s, over_d = (cnt_s > N).tolist(), (cnt_d > N).tolist()
    if any(over_s) or any(over_d):
        cs, cd = cnt_s.tolist(), cnt_d.tolist()
        ss, sd = st.start[st.find(pairs[:, 0])].tolist(), dt.start[dt.find(pairs[:, 1])].tolist()
        for k in range(len(pairs)):
            if over_s[k]:
                rows_s[k] = ss[k] + torch.randperm(cs[k])[0:N].to(dev)
            if over_d[k]:
                rows_d[k] = sd[k] + torch.randperm(cd[k])[0:N].to(dev)
    B = len(pairs)
    segs = torch.empty((2, B, N, 4), dtype=torch.float32, device=dev)
    for which, (table, rows) in enumerate(((st, rows_s), (dt, rows_d))):
        src_rows = torch.where(rows >= 0, table.order[rows.clamp(min=0)], rows).to(torch.int32).contiguous()
        _lib.call("icpflow_gather_pad", _lib.ptr(table.points), _lib.ptr(src_rows), B, N, _lib.ptr(segs[which]),
                  _lib.stream(dev))
    segs_src, segs_dst = segs[0], segs[1]
    transformations = hist_icp(args, segs_src, segs_dst)
    errors, inliers, ratios, ious, translations, rotations = match_eval(args, segs_src, segs_dst, transformations)
    keep = check_transformation(args, translations, rotations, ious.min(dim=1)[0])
    S, D = len(st.labels_unq), len(dt.labels_unq)
    m_err = torch.full((S, D, 2), 1e8, device=dev)
    m_inl = torch.zeros((S, D, 2), device=dev)
    m_rat = torch.zeros((S, D, 2), device=dev)
    m_iou = torch.zeros((S, D, 2), device=dev)
    m_T = torch.zeros((S, D, 4, 4), device=dev)
    si, di = st.find(pairs[:, 0])[keep], dt.find(pairs[:, 1])[keep]
    if len(si) == 0:
        return torch.zeros((0, 10), device=dev), torch.zeros((0, 4, 4), device=dev)
    m_err[si, di] = errors[keep]
    m_inl[si, di] = inliers[keep]
    m_rat[si, di] = ratios[keep]
    m_iou[si, di] = ious[keep]
    m_T[si, di] = transformations[keep]
    err_min = m_err.min(-1)[0]
    rows = torch.arange(S, device=dev)
    best = torch.argmin(err_min, dim=1)                                   # utils_helper.py:108-110
    valid = err_min[rows, best] < args.thres_error                       # utils_match.py:112
    rows, best = rows[valid], best[valid]
    out = torch.cat([st.labels_unq[rows][:, None], dt.labels_unq[best][:, None], m_err[rows, best],
                     m_inl[rows, best], m_rat[rows, best], m_iou[rows, best]], dim=1)
    return out, m_T[rows, best]


def setdiff1d(t1, t2):
    """utils_helper.py:172-183: labels of t1 not in t2 (t2 a subset of t1), sorted."""
    t12, counts = torch.cat([torch.unique(t1), torch.unique(t2)]).unique(return_counts=True)
    return t12[counts == 1]


def match_pcds(args, src_points, dst_points, src_labels, dst_labels):
    """utils_match.py:24-66: stage 1 registers clusters that keep their label across the two
    frames (static / slow objects), stage 2 every remaining source cluster against every remaining
    destination cluster.  -> pairs [P,10] (labels, errors, inliers, ratios, ious), transforms [P,4,4]."""
    _lib.require_gpu(src_points, dst_points, src_labels, dst_labels)
    dev = src_points.device
    st, dt = ClusterTable(src_points, src_labels), ClusterTable(dst_points, dst_labels)
    src_unq, dst_unq = st.labels_unq.long(), dt.labels_unq.long()
    labels_unq = torch.unique(torch.cat([src_unq, dst_unq]))
    empty = (torch.zeros((0, 10), device=dev), torch.zeros((0, 4, 4), device=dev))

    pairs = torch.stack([labels_unq, labels_unq], dim=1)
    pairs = pairs[pairs.min(dim=1)[0] >= 0]
    pairs_true = sanity_check(args, st, dt, pairs)
    pairs_sta, T_sta = match_pairs(args, src_points, dst_points, src_labels, dst_labels, pairs_true, (st, dt)) \
        if len(pairs_true) > 0 else empty

    if len(pairs_sta) < len(labels_unq):
        if len(pairs_sta) > 0:
            src_unq = setdiff1d(src_unq, pairs_sta[:, 0].long())
            dst_unq = setdiff1d(dst_unq, pairs_sta[:, 1].long())
        pairs = torch.stack([src_unq.repeat_interleave(len(dst_unq)), dst_unq.repeat(len(src_unq))], dim=1)
        pairs_true = sanity_check(args, st, dt, pairs)
    else:
        pairs_true = pairs[:0]
    pairs_dyn, T_dyn = match_pairs(args, src_points, dst_points, src_labels, dst_labels, pairs_true, (st, dt)) \
        if len(pairs_true) > 0 else empty
    return torch.cat([pairs_sta, pairs_dyn], dim=0), torch.cat([T_sta, T_dyn], dim=0)
