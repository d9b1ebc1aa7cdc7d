"""Drop-ins for the registration entry points of the reference's utils_match.py."""
import ctypes

import numpy as np
import torch

from . import _lib
from .utils_check import ClusterTable, _sanity_mask, check_transformation, sanity_check, sanity_grid
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
    iters = torch.empty((1,), dtype=torch.int32, device=s.device)   # always written by the call
    _lib.check_vote_bins(B, lens)
    ws = _lib.workspace(s.device, _lib.workspace_bytes(B, N, lens))
    _lib.call("icpflow_hist_icp", _lib.ptr(s), _lib.ptr(d), B, N, _lib.ptr(ex), lens[0], _lib.ptr(ey),
              lens[1], _lib.ptr(ez), lens[2], float(args.thres_dist // 2), float(args.thres_dist), max_it,
              rel, stop, _lib.ptr(out), _lib.ptr(iters), _lib.ptr(ws), ws.numel(), _lib.stream(s.device), _lib.opt())
    return (out, iters) if return_iterations else out


def hist_icp_many(args, srcs, dsts, return_iterations=False):
    """K independent hist_icp batches (same max_points) in flight in ONE call (icpflow_hist_icp_many): each batch keeps
    its own batch-global ICP stop and returns exactly what `hist_icp` returns for it; the library overlaps the tail of
    one batch's ICP launch with the vote and scoring of the others on internal streams forked from / joined into the
    current stream.  srcs, dsts: sequences of float32 [B_k, max_points, 4] -> list of [B_k,4,4] (and of iteration counts)."""
    K = len(srcs)
    assert K == len(dsts) and K > 0
    ss = [_lib.cloud(x, "src") for x in srcs]
    dd = [_lib.cloud(x, "dst") for x in dsts]
    N = ss[0].shape[1]
    dev = ss[0].device
    assert all(a.shape == b.shape and a.shape[1] == N and a.device == dev for a, b in zip(ss, dd))
    ex, ey, ez = bin_edges(args, dev)
    lens = (len(ex), len(ey), len(ez))
    max_it, rel, stop = _icp_options(args)
    outs = [torch.empty((a.shape[0], 4, 4), dtype=torch.float32, device=dev) for a in ss]
    iters = [torch.empty((1,), dtype=torch.int32, device=dev) for _ in ss]
    if _lib._current()[-1]["vote_bins"] is not None:
        raise RuntimeError("options(vote_bins=...) describes ONE batch: not available through hist_icp_many")
    need = [_lib.workspace_bytes(a.shape[0], N, lens) for a in ss]
    ws = _lib.workspaces(dev, need)
    arr = lambda ts: (ctypes.c_void_p * K)(*[t.data_ptr() for t in ts])   # noqa: E731
    _lib.call("icpflow_hist_icp_many", K, arr(ss), arr(dd), (ctypes.c_int * K)(*[a.shape[0] for a in ss]), N,
              _lib.ptr(ex), lens[0], _lib.ptr(ey), lens[1], _lib.ptr(ez), lens[2], float(args.thres_dist // 2),
              float(args.thres_dist), max_it, rel, stop, arr(outs), arr(iters), arr(ws),
              (ctypes.c_size_t * K)(*[w.numel() for w in ws]), _lib.stream(dev), _lib.opt())
    return (outs, iters) if return_iterations else outs


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
              _lib.ptr(o3[1]), _lib.ptr(ws), ws.numel(), _lib.stream(dev), _lib.opt())
    return o2[0], o2[1], o2[2], o2[3], o3[0], o3[1]


# layout of the flat result buffer of one hist_icp_eval call: float32 offsets in units of B (transforms 16, errors,
# inliers, ratios, ious 2 each, translations, rotations 3 each), then ONE int32: the iteration count
_EVAL_COLS = (0, 16, 18, 20, 22, 24, 27, 30)


def _hist_icp_eval_flat(args, src, dst):
    """icpflow_hist_icp_eval with every result in ONE float32 buffer [30 B + 1] (array after array, the iteration count
    as the last word): one allocation, and one transfer for a caller that wants the numbers on the host."""
    s = _lib.cloud(src, "src")
    d = _lib.cloud(dst, "dst")
    assert s.shape == d.shape, "src and dst must share [B, max_points, 4]"
    B, N, _ = s.shape
    dev = s.device
    ex, ey, ez = bin_edges(args, dev)
    lens = (len(ex), len(ey), len(ez))
    max_it, rel, stop = _icp_options(args)
    flat = torch.empty((30 * B + 1,), dtype=torch.float32, device=dev)
    base = flat.data_ptr()
    at = [ctypes.c_void_p(base + 4 * B * c) for c in _EVAL_COLS]
    _lib.check_vote_bins(B, lens)
    ws = _lib.workspace(dev, _lib.workspace_bytes(B, N, lens))
    _lib.call("icpflow_hist_icp_eval", _lib.ptr(s), _lib.ptr(d), B, N, _lib.ptr(ex), lens[0], _lib.ptr(ey),
              lens[1], _lib.ptr(ez), lens[2], float(args.thres_dist // 2), float(args.thres_dist), max_it,
              rel, stop, at[0], at[7], at[1], at[2], at[3], at[4], at[5], at[6], _lib.ptr(ws), ws.numel(),
              _lib.stream(dev), _lib.opt())
    return flat, B


def _eval_views(r, B):
    """The arrays of a flat result buffer (device tensor or host numpy array): T [B,4,4], the six metric arrays, iterations [1]."""
    c = _EVAL_COLS
    part = [r[B * c[k]: B * c[k + 1]] for k in range(7)]
    shapes = [(B, 4, 4), (B, 2), (B, 2), (B, 2), (B, 2), (B, 3), (B, 3)]
    arrs = [x.reshape(sh) for x, sh in zip(part, shapes)]
    iters = r[30 * B:].view(torch.int32 if isinstance(r, torch.Tensor) else np.int32)
    return arrs[0], tuple(arrs[1:]), iters


def hist_icp_eval(args, src, dst, return_iterations=False):
    """`hist_icp(args, src, dst)` and `match_eval(args, src, dst, T)` of its result in ONE call (icpflow_hist_icp_eval):
    what match_pairs does for every batch of candidate pairs (utils_match.py:92-93).  Same numbers as the two calls;
    the metrics reuse the valid-row counts and the sorted clouds the registration leaves in its workspace.
    -> (T [B,4,4], (errors, inliers, ratios, ious [B,2], translations, rotations [B,3])[, iterations])."""
    flat, B = _hist_icp_eval_flat(args, src, dst)
    out, ev, iters = _eval_views(flat, B)
    return (out, ev, iters) if return_iterations else (out, ev)


# --------------------------------------------------------------------------------------
# the caller of the registration path: cluster association
#
# Device work per association stage: one gather of the candidate clusters into the padded batch,
# hist_icp, match_eval.  Everything else -- candidate lists, sanity_check, the reject test, the S x D
# matrices and the row arg-min -- involves a few hundred numbers and runs on the host copy of the
# cluster tables (utils_check.ClusterTable) in numpy; a stage costs ONE device -> host transfer of the
# 30 B + 1 result words instead of the reference's per-pair scalar reads.
# --------------------------------------------------------------------------------------
class _Staging:
    """Pinned host buffers that kernels read IN PLACE (per stream and slot): a stage's few KB of segment rows, subsamples and
    candidate rows are written here by the host and never uploaded -- the gather and assignment kernels load them over PCIe
    (tools/dbg/zero_copy_probe.py: 5 us of host time and 6 us to the end of the gather, against 26 / 26 us with a pageable
    .to() in front and 9 / 21 us with a pinned non-blocking copy).  The host writes without regard to the stream, so a slot
    carries an event: `done()` records it behind the launches that read the buffer, `get()` waits for it before handing the
    buffer out again (a stage's results are normally read back long before: the wait is a query)."""
    _slots = {}

    @staticmethod
    def get(dev, slot, nbytes):
        """-> (key, uint8 numpy view of nbytes, address for the kernels)."""
        key = (_lib.stream_handle(dev), slot)
        ent = _Staging._slots.get(key)
        if ent is None or ent[0].numel() < nbytes:
            if ent is not None:
                ent[1].synchronize()
            ent = (torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
            _Staging._slots[key] = ent
        else:
            ent[1].synchronize()
        return key, ent[0].numpy()[:nbytes], ent[0].data_ptr()

    @staticmethod
    def done(key):
        _Staging._slots[key][1].record()


def _stage_rows(args, st, dt, si, di):
    """The segment rows of a stage's candidate clusters (rows si / di of the tables) for icpflow_gather_segments, in a
    staging buffer the kernels read in place (_Staging; the caller calls _Staging.done(key) behind its launches):
    -> (key, address of int64 [2,3,B] rows (start, length, offset of the subsample or -1) followed by the int32 subsamples, B,
    width, whether there are subsamples).  Over-long clusters are subsampled with
    torch.randperm on the host generator (or `args.generator`, a torch.Generator private to the caller, so that
    frame pairs registered concurrently do not share a stream of draws), src then dst, pair by pair -- the
    reference's stream of draws
    (utils_match.py:84-89, utils_helper.py:198-201)."""
    dev = st.points.device
    B = len(si)
    cs, cd = st.h_count[si], dt.h_count[di]
    # Padded to the longest cluster of THIS batch (in steps of 64 rows), max_points at most: padding rows carry flag 0 and
    # no entry point reads them, so the registrations are those of the max_points-wide batch (the sums of a pair follow
    # the shape of the workgroups that serve it: agreement to rounding, not to the bit) -- but every sort, vote and scan
    # of the stage is sized by the width.  Stage 2 of a frame pair holds the small clusters stage 1 left over: 12 pairs of
    # <= 540 points at max_points 10000 on the demo frame.
    N = int(args.max_points)
    cap = N                                                  # (clusters above it are subsampled to exactly max_points)
    N = min(N, max(64, (int(max(cs.max(), cd.max())) + 63) // 64 * 64)) if getattr(args, "tight_padding", True) else N
    over = np.nonzero((cs > cap) | (cd > cap))[0]
    n_perm = int((cs[over] > cap).sum() + (cd[over] > cap).sum())
    # the int64 segment rows [2, 3, B] (start, length, offset of the subsample or -1), then the int32 subsamples
    key, host, address = _Staging.get(dev, "rows", 48 * B + 4 * cap * n_perm)
    seg = host[: 48 * B].view(np.int64).reshape(2, 3, B)
    perm = host[48 * B:].view(np.int32)
    seg[0, 0], seg[0, 1], seg[0, 2] = st.h_start[si], np.minimum(cs, cap), -1
    seg[1, 0], seg[1, 1], seg[1, 2] = dt.h_start[di], np.minimum(cd, cap), -1
    drawn = 0
    for k in over:                                          # random_choice, utils_helper.py:198-201
        for which, c in ((0, cs), (1, cd)):
            if c[k] > cap:
                seg[which, 2, k] = drawn * cap
                perm[drawn * cap: (drawn + 1) * cap] = torch.randperm(int(c[k]), generator=getattr(args, "generator", None))[0:cap].numpy()
                drawn += 1
    return key, address, B, N, n_perm > 0


def _gather_pair_batches(args, st, dt, si, di):
    """pad_segment (utils_helper.py:185-196) of the candidate clusters (rows si / di of the tables), one
    kernel per cloud: -> two [B, width, 4] device tensors (width: _stage_rows)."""
    dev = st.points.device
    key, base, B, N, has_perm = _stage_rows(args, st, dt, si, di)
    d_perm = ctypes.c_void_p(base + 48 * B) if has_perm else None
    segs = torch.empty((2, B, N, 4), dtype=torch.float32, device=dev)
    for which, table in enumerate((st, dt)):
        _lib.call("icpflow_gather_segments", _lib.ptr(table.points), _lib.ptr(table.order), ctypes.c_void_p(base + 24 * B * which),
                  d_perm, B, N, _lib.ptr(segs[which]), _lib.stream(dev))
    _Staging.done(key)
    return segs[0], segs[1]


def _registration(args, dev):
    """icpflow_registration_t of `args` (+ the tensors it points to, to be kept alive by the caller)."""
    ex, ey, ez = bin_edges(args, dev)
    max_it, rel, stop = _icp_options(args)
    reg = _lib.Registration(ex.data_ptr(), ey.data_ptr(), ez.data_ptr(), len(ex), len(ey), len(ez), float(args.thres_dist // 2),
                            float(args.thres_dist), float(rel), int(max_it), int(stop))
    return reg, (ex, ey, ez)


def _launch_pairs(args, st, dt, pairs):
    """The device half of utils_match.py:69-136 for numpy candidate `pairs` [K,2]: gather, hist_icp, match_eval, all
    asynchronous.  -> what _finish_pairs needs (host work placed between the two overlaps the kernels)."""
    si, di = st.find_host(pairs[:, 0]), dt.find_host(pairs[:, 1])
    assert (si >= 0).all() and (di >= 0).all()
    r = _register_stage(args, st, dt, si, di)[0]
    return si, di, r


def _register_stage(args, st, dt, si, di, d_si=None, d_di=None):
    """One association stage in ONE call into the library (icpflow_register_stage: both padded clouds, hist_icp, match_eval).
    -> (flat result tensor [30 K + 1], the icpflow_stage_t, what it points to)."""
    dev = st.points.device
    key, base, K, N, has_perm = _stage_rows(args, st, dt, si, di)
    reg, edges = _registration(args, dev)
    lens = (reg.len_x, reg.len_y, reg.len_z)
    _lib.check_vote_bins(K, lens)
    tables = _lib.Tables(st.points.data_ptr(), st.order.data_ptr(), st._packed.data_ptr() + 8, dt.points.data_ptr(),
                         dt.order.data_ptr(), dt._packed.data_ptr() + 8, len(st.h_labels), len(dt.h_labels), 9)
    n_clouds = 2 * K * N * 4
    scratch = torch.empty((n_clouds + 30 * K + 1,), dtype=torch.float32, device=dev)
    stage = _lib.Stage(base, base + 48 * K if has_perm else None, d_si, d_di, scratch.data_ptr(), scratch.data_ptr() + 4 * n_clouds, K, N)
    ws = _lib.workspace(dev, _lib.workspace_bytes(K, N, lens))
    # The stages of a frame pair keep their teams of workgroups on half of the GPU (ICPFLOW_OPT_TEAMS_HALF_GPU): a frame's
    # stage needs far fewer workgroups than the GPU has CUs, and two frame pairs in flight can then run their team launches
    # side by side instead of one after the other.  Always, in flight or not: the plan decides the order of a team's sums, and
    # a frame pair registers to the same bits whatever else is in flight (`args.teams_full_gpu = True`: the full-GPU plan).
    with _lib.options(teams_half_gpu=not getattr(args, "teams_full_gpu", False), no_shared_scans=not getattr(args, "shared_scans", False)):
        _lib.call("icpflow_register_stage", ctypes.byref(tables), ctypes.byref(stage), ctypes.byref(reg), _lib.ptr(ws), ws.numel(),
                  _lib.stream(dev), _lib.opt())
    _Staging.done(key)
    return scratch[n_clouds:], stage, (tables, reg, lens, scratch, edges)


class Pending:
    """A device tensor on its way to the host.  Synchronous form (the default): `.cpu()` right away.  Asynchronous form
    (frame pairs in flight, frame_pairs.run_stream): a copy into pinned memory on the current stream + an event; the
    caller's scheduler resumes the frame pair once `ready()`, `get()` then returns without blocking."""
    _pinned = {}

    def __init__(self, t, asynchronous=False, slot=0):
        if not asynchronous:
            self.host, self.ev, self._dev = None, None, t      # (copied when asked for: host work in between overlaps the GPU)
            return
        key = (_lib.stream_handle(t.device), slot, t.dtype)
        buf = Pending._pinned.get(key)
        if buf is None or buf.numel() < t.numel():
            buf = torch.empty(max(t.numel(), 1 << 14), dtype=t.dtype, pin_memory=True)
            Pending._pinned[key] = buf
        self.host = buf[: t.numel()].view(t.shape)
        self.host.copy_(t, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()

    def ready(self):
        return self.ev is None or self.ev.query()

    def get(self):
        if self.ev is not None:
            self.ev.synchronize()
        elif self.host is None:
            self.host = self._dev.cpu()
        return self.host.numpy()


def _match_pairs_host(args, st, dt, pairs):
    """utils_match.py:69-136 on numpy candidate `pairs` [K,2] -> (pairs [P,10], transforms [P,4,4]) numpy."""
    si, di, r = _launch_pairs(args, st, dt, pairs)
    return _finish_pairs(args, st, dt, (si, di, r.cpu().numpy()))


def _finish_pairs(args, st, dt, launched):
    """The host half, on the results of the stage brought to the host in ONE transfer (the flat buffer of
    _hist_icp_eval_flat): reject test, S x D matrices, row arg-min."""
    si, di, r = launched
    B = len(si)
    T_h, (errors, inliers, ratios, ious, translations, rotations), iters = _eval_views(r, B)
    if iters[0] < 0:
        # a team of workgroups sharing one large pair gave up waiting for a member (include/icpflow_hip.h, a-5):
        # the transforms are NaN.  Never let that pass as "no match" -- the points would silently get ego flow only.
        raise RuntimeError("icpflow_hist_icp abandoned the batch: a wait between workgroups timed out -- a team sharing one "
                           "large pair, or a helper's hand-off in a persistent launch (GPU shared with another process?); "
                           "retry, or register with _lib.options(no_teams=True, no_helpers=True, no_persistent=True)")
    # (Python's builtin min(iou), utils_match.py:99: iou[1] if iou[1] < iou[0] else iou[0]; differs from np.minimum only on NaNs)
    keep = check_transformation(args, translations, rotations, np.where(ious[:, 1] < ious[:, 0], ious[:, 1], ious[:, 0]))
    if not keep.any():
        return np.zeros((0, 10), np.float32), np.zeros((0, 4, 4), np.float32)
    # The reference fills S x D matrices (1e8 where no candidate), takes the row arg-min of min(err_src, err_dst) -- the FIRST
    # minimal column -- and keeps the rows below thres_error (utils_helper.py:108-110, utils_match.py:112).  The same choice on
    # the list of kept candidates (a few dozen rows instead of S x D entries; the (source, destination) pairs of a stage are
    # distinct): per source row the smallest error, ties to the smallest destination index; a row whose best error is not below
    # thres_error (< 1e8) has no match, and rows come out in ascending source order like the matrix rows.
    ks = np.nonzero(keep)[0]
    err = np.minimum(errors[ks, 0], errors[ks, 1])
    if np.isnan(err).any():
        # np.argmin takes a NaN for the minimum of its row, and NaN < thres_error is False: such a row has no match
        ok = ~np.isin(si[ks], si[ks][np.isnan(err)])
        ks, err = ks[ok], err[ok]
    order = np.lexsort((di[ks], err, si[ks]))
    ks, err = ks[order], err[order]
    first = np.ones(len(ks), dtype=bool)
    first[1:] = si[ks[1:]] != si[ks[:-1]]
    k = ks[first & (err < np.float32(args.thres_error))]
    out = np.concatenate([st.h_labels[si[k]][:, None], dt.h_labels[di[k]][:, None], errors[k], inliers[k], ratios[k],
                          ious[k]], axis=1).astype(np.float32)
    return out, T_h[k].astype(np.float32)


def match_pairs(args, src_points, dst_points, src_labels, dst_labels, pairs, tables=None):
    """utils_match.py:69-136: register every candidate pair, reject implausible transforms
    (check_transformation), assign each source cluster its best destination cluster (row arg-min
    of min(err_src, err_dst) below thres_error).  -> pairs [P,10], transformations [P,4,4] (device)."""
    assert len(pairs) > 0
    dev = src_points.device
    st, dt = tables if tables is not None else ClusterTable.pair(src_points, src_labels, dst_points, dst_labels)
    p = pairs.detach().cpu().numpy() if isinstance(pairs, torch.Tensor) else np.asarray(pairs)
    out, T = _match_pairs_host(args, st, dt, p.astype(np.float32))
    return torch.from_numpy(out).to(dev), torch.from_numpy(T).to(dev)


def setdiff1d(t1, t2):
    """utils_helper.py:172-183: labels of t1 not in t2 (t2 a subset of t1), sorted; numpy or tensors."""
    if isinstance(t1, torch.Tensor):
        t12, counts = torch.cat([torch.unique(t1), torch.unique(t2)]).unique(return_counts=True)
        return t12[counts == 1]
    # (the labels counted once in unique(t1) ++ unique(t2), like the tensor branch and the reference: with t2 a subset of
    # t1 -- the only way match_pcds calls it -- those are the labels of t1 alone; a label of t2 alone is kept too)
    t1, t2 = np.asarray(t1), np.asarray(t2)
    u1 = t1 if (t1.ndim == 1 and (len(t1) < 2 or (t1[1:] > t1[:-1]).all())) else np.unique(t1)   # (match_pcds passes sorted unique labels)
    if len(t2) == 0:
        return u1
    if len(u1):
        pos = np.minimum(np.searchsorted(u1, t2), len(u1) - 1)
        if (u1[pos] == t2).all():                       # the subset case, the only one match_pcds produces: one scatter
            keep = np.ones(len(u1), dtype=bool)
            keep[pos] = False
            return u1[keep]
    u2 = np.unique(t2)
    return np.concatenate([u1[~np.isin(u1, u2)], u2[~np.isin(u2, u1)]])


def match_pcds_steps(args, src_points, dst_points, src_labels, dst_labels, asynchronous=False):
    """utils_match.py:24-66 as a generator: yields a `Pending` at each of its three device -> host hand-overs (cluster
    tables, results of stage 1, results of stage 2) and returns (pairs [P,10], transforms [P,4,4]) device tensors.
    `match_pcds` drives it to the end; frame_pairs.run_stream keeps several frame pairs in flight and resumes each one
    when its transfer has landed, so that the host half of one frame pair runs under the kernels of the others."""
    _lib.require_gpu(src_points, dst_points, src_labels, dst_labels)
    dev = src_points.device
    if len(src_points) == 0 or len(dst_points) == 0:
        # a frame whose points were all filtered out: no cluster, no candidate pair -- the reference's match_pcds returns
        # empty pairs for it (utils_match.py:24-66 with empty label sets); the cluster-table kernels want rows
        return torch.zeros((0, 10), dtype=torch.float32, device=dev), torch.zeros((0, 4, 4), dtype=torch.float32, device=dev)
    st, dt = ClusterTable.pair(src_points, src_labels, dst_points, dst_labels, fetch=False)
    pend = Pending(st._both, asynchronous, 0)
    yield pend
    both = pend.get()
    st.fetch(both[0])
    dt.fetch(both[1])
    src_unq, dst_unq = st.h_labels.astype(np.int64), dt.h_labels.astype(np.int64)
    labels_unq = np.unique(np.concatenate([src_unq, dst_unq]))
    empty = (np.zeros((0, 10), np.float32), np.zeros((0, 4, 4), np.float32))

    pairs = np.stack([labels_unq, labels_unq], axis=1)
    pairs = pairs[np.minimum(pairs[:, 0], pairs[:, 1]) >= 0].astype(np.float32)                  # :30-31
    pairs_true = pairs[_sanity_mask(args, st, dt, pairs)] if len(pairs) else pairs
    if len(pairs_true) > 0 and _device_association_ok(args, st, dt, asynchronous):
        # both stages, the step between them, the pair rows: enqueued without reading a stage's results back (below)
        out = yield from _match_pcds_device(args, st, dt, pairs_true, asynchronous)
        args.association_path = "device" if out is not None else "host"
        if out is not None:
            return out
        # (a cluster too long for max_points turned out to need a second try: its random subsample must be drawn in the
        # reference's order, which only the path below knows -- registered again from the start, same generator state)
    launched = _launch_pairs(args, st, dt, pairs_true) if len(pairs_true) > 0 else None
    pend = Pending(launched[2], asynchronous, 1) if launched is not None else None
    # while stage 1 runs on the GPU: the sanity test of EVERY source cluster against every destination cluster (stage 2
    # reads the rows and columns of the clusters stage 1 leaves unmatched)
    grid = sanity_grid(args, st, dt, np.arange(len(src_unq)), np.arange(len(dst_unq))) if len(src_unq) and len(dst_unq) else None
    if pend is not None:
        yield pend
        pairs_sta, T_sta = _finish_pairs(args, st, dt, (launched[0], launched[1], pend.get()))
    else:
        pairs_sta, T_sta = empty

    if len(pairs_sta) < len(labels_unq):                                                          # :42
        if len(pairs_sta) > 0:
            src_unq = setdiff1d(src_unq, pairs_sta[:, 0].astype(np.int64))
            dst_unq = setdiff1d(dst_unq, pairs_sta[:, 1].astype(np.int64))
        # every remaining source against every remaining destination (:45-53), tested on the S x D grid;
        # surviving candidates in the reference's order (source-major)
        si, di = st.find_host(src_unq.astype(np.float32)), dt.find_host(dst_unq.astype(np.float32))
        rs, rd = np.nonzero(grid[np.ix_(si, di)]) if len(si) and len(di) else (si[:0], di[:0])
        pairs_true = np.stack([src_unq[rs], dst_unq[rd]], axis=1).astype(np.float32).reshape(-1, 2)
    else:
        pairs_true = pairs[:0]
    if len(pairs_true) > 0:
        launched = _launch_pairs(args, st, dt, pairs_true)
        pend = Pending(launched[2], asynchronous, 2)
        yield pend
        pairs_dyn, T_dyn = _finish_pairs(args, st, dt, (launched[0], launched[1], pend.get()))
    else:
        pairs_dyn, T_dyn = empty
    P = len(pairs_sta) + len(pairs_dyn)
    both = np.empty((26 * P,), dtype=np.float32)             # ONE upload: the pair rows [P,10], then the transforms [P,4,4]
    np.concatenate([pairs_sta, pairs_dyn], axis=0, out=both[: 10 * P].reshape(P, 10))
    np.concatenate([T_sta, T_dyn], axis=0, out=both[10 * P:].reshape(P, 4, 4))
    d_both = torch.from_numpy(both).to(dev, non_blocking=asynchronous)
    return d_both[: 10 * P].view(P, 10), d_both[10 * P:].view(P, 4, 4)


def _device_association_ok(args, st, dt, asynchronous=False):
    """The device-side association (icpflow_assoc_assign / _collect, options.d_pair_active) needs the ICP's single speculative
    launch (reference stop rule, <= 128 iterations) and cluster tables that fit the kernels' LDS tables.
    `args.device_association`: True / False (the host path: every stage read back, numpy in between), default None = on for a
    frame pair registered on its own (`asynchronous` False), off for frame pairs in flight.  On its own a frame pair gains the
    read-back between the stages, time the GPU idled (2.15-2.24 -> 2.01-2.03 ms at 2048 points, 2.33 -> 2.21-2.26 at 10000);
    a stream of frame pairs in flight is bound by its ONE host thread (tools/dbg/stream_host_busy.py: wall = host busy + 0.08
    ms), the read-backs were hidden behind the other frame pairs already, and the device path costs that thread more (the
    superset; calls that block longer the more is queued on the GPU: 1.25-1.64 ms busy per frame pair against 1.10-1.31).
    The two paths agree on the matched pairs and to rounding on the numbers (stage 2's batch has another shape:
    tests/test_gpu_parity.py::test_device_association_equals_the_host_path, 1e-5 m on the flow); with the option set either
    way a frame pair registers to the same bits in flight and on its own."""
    want = getattr(args, "device_association", None)
    if want is None:
        want = not asynchronous
    if not want:
        return False
    max_it, _, stop = _icp_options(args)
    cur = _lib._current()[-1]
    return (stop == 0 and 2 <= max_it <= 128 and cur["arith"] == 0 and not (cur["flags"] & _lib.OPT_FLAGS["no_speculative"])
            and len(st.h_labels) <= 1024 and len(dt.h_labels) <= 1024)


def _match_pcds_device(args, st, dt, pairs_true, asynchronous):
    """match_pcds from the cluster tables on, with ONE hand-over at the end.  Stage 1 as usual; stage 2 is
    enqueued for a SUPERSET of its candidates -- every (source, destination) the sanity grid lets through, known from the tables
    alone -- and a one-workgroup kernel (icpflow_assoc_assign) does on the device what _finish_pairs does on the host, then
    switches every stage-2 candidate on or off (both clusters still without a partner, utils_match.py:45-53): the switched-off
    ones are handed over as empty clouds and flagged in options.d_pair_active, which keeps them out of the ICP's batch-global
    stop -- stage 2's batch IS the reference's batch.  icpflow_assoc_collect writes the pair rows of both stages (two calls into the library:
    icpflow_register_stage for stage 1, icpflow_associate_frame for everything behind it).
    -> (pairs [P,10], transforms [P,4,4]) device tensors, or None when a candidate of stage 2 that had to be left out of the
    superset -- a cluster longer than max_points, whose random subsample must be drawn in the reference's order of draws --
    turned out to be needed: the caller then runs the host path."""
    dev = st.points.device
    S, D = len(st.h_labels), len(dt.h_labels)
    gen_state = None
    g = getattr(args, "generator", None)
    gen_state = g.get_state() if g is not None else torch.get_rng_state()
    si1, di1 = st.find_host(pairs_true[:, 0]), dt.find_host(pairs_true[:, 1])
    K1 = len(si1)
    # stage 1 first (ONE call: both padded clouds, hist_icp, match_eval): the host work below (grid, superset) runs while the
    # GPU is in it
    _, stage1, (tables, reg, lens, *keep_alive) = _register_stage(args, st, dt, si1, di1)
    ws_bytes1 = _lib.workspace_bytes(K1, stage1.N, lens)
    # Left out of the superset: pairs with a cluster longer than max_points (its random subsample must be drawn in the
    # reference's order of draws, which depends on what stage 1 matches) -- and pairs with a cluster longer than
    # `args.device_association_width` (1024): the superset's batch is as wide as its longest cluster, and a wide batch of two
    # hundred mostly switched-off pairs reserves LDS for its width on every CU it touches (130 KiB at 2944 points: with frame
    # pairs in flight its workgroups queued behind the other frames' team launches and held up everything behind them, 1.4
    # -> 2.1 ms per frame pair).  Long clusters are the ones stage 1 matches; if one of them does need its second try, the
    # check at the end sends the frame pair through the host path.
    cap_pts = min(int(args.max_points), int(getattr(args, "device_association_width", 1024)))
    grid = sanity_grid(args, st, dt, np.arange(S), np.arange(D))
    long_s, long_d = st.h_count > cap_pts, dt.h_count > cap_pts
    left_out = grid & (long_s[:, None] | long_d[None, :])
    rs, rd = np.nonzero(grid & ~left_out)
    K2 = len(rs)
    N2 = min(cap_pts, max(64, (int(max(st.h_count[rs].max(), dt.h_count[rd].max())) + 63) // 64 * 64)) if K2 else 64
    if not getattr(args, "tight_padding", True):
        N2 = int(args.max_points)
    # stage 2's segment rows [2,3,K2] int64, then the candidate rows of both stages as int32: read in place by the kernels
    # (_Staging; a pageable upload here waited for everything queued on its stream -- the host sat out stage 1 and frame pairs
    # in flight stopped overlapping, 1.3 -> 1.9 ms per frame pair)
    nbytes = 48 * K2 + 8 * (K1 + K2)
    key2, host, base = _Staging.get(dev, "superset", nbytes)
    seg2 = host[: 48 * K2].view(np.int64).reshape(2, 3, K2)
    idx = host[48 * K2:].view(np.int32)
    if K2:
        seg2[0, 0], seg2[0, 1], seg2[0, 2] = st.h_start[rs], st.h_count[rs], -1
        seg2[1, 0], seg2[1, 1], seg2[1, 2] = dt.h_start[rd], dt.h_count[rd], -1
    idx[0:K1], idx[K1:2 * K1] = si1, di1
    idx[2 * K1:2 * K1 + K2], idx[2 * K1 + K2:] = rs, rd
    stage1.d_si, stage1.d_di = base + 48 * K2, base + 48 * K2 + 4 * K1
    # everything behind stage 1's registration in ONE call: the assignment of stage 1 (which switches stage 2's candidates on
    # or off), stage 2, its assignment, the pair rows of both -- and the flow of the frame pair when the caller asked for it
    # (frame_pairs: `args.flow_request`; rows beyond the matches carry a label no point has and the identity, so the padded
    # pair rows serve and nothing has to be read back first)
    cap = 2 * S
    n_clouds2, n_res2 = 2 * K2 * N2 * 4, 30 * K2 + 1
    n_small = 2 * S + 2
    scratch2 = torch.empty((n_clouds2 + n_res2 + n_small + 26 * cap + (K2 + 3) // 4,), dtype=torch.float32, device=dev)
    o = scratch2.data_ptr()
    stage2 = _lib.Stage(base, None, base + 48 * K2 + 8 * K1, base + 48 * K2 + 8 * K1 + 4 * K2, o, o + 4 * n_clouds2, K2, N2)
    o_small = n_clouds2 + n_res2
    small = scratch2[o_small: o_small + n_small].view(torch.int32)
    rows = scratch2[o_small + n_small: o_small + n_small + 10 * cap].view(cap, 10)
    T = scratch2[o_small + n_small + 10 * cap: o_small + n_small + 26 * cap].view(cap, 4, 4)
    p_active2 = o + 4 * (o_small + n_small + 26 * cap)
    f32 = lambda v: float(np.float32(v))   # noqa: E731
    req = getattr(args, "flow_request", None)
    flow = f_pts = f_lab = f_pose = None
    if req is not None:
        f_pts, f_lab, f_pose = (req["points"][:, 0:3].contiguous().float(), req["labels"].contiguous().float(),
                                req["pose"].to(dev).contiguous().float())
        assert len(f_pts) == len(f_lab)
        flow = torch.empty((len(f_pts), 3), dtype=torch.float32, device=dev)
    if K2:
        _lib.check_vote_bins(K2, lens)
    ws = _lib.workspace(dev, max(ws_bytes1, _lib.workspace_bytes(K2, N2, lens) if K2 else 0))
    with _lib.options(teams_half_gpu=not getattr(args, "teams_full_gpu", False), no_shared_scans=not getattr(args, "shared_scans", False)):
        _lib.call("icpflow_associate_frame", ctypes.byref(tables), ctypes.byref(stage1), ctypes.byref(stage2) if K2 else None,
                  ctypes.c_void_p(p_active2) if K2 else None, ctypes.byref(reg), f32(args.translation_frame), f32(args.thres_iou),
                  f32(args.thres_rot * 90.0), f32(args.thres_error), _lib.ptr(small), cap, _lib.ptr(rows), _lib.ptr(T),
                  _lib.ptr(f_pts), _lib.ptr(f_lab), len(f_pts) if flow is not None else 0, _lib.ptr(f_pose), _lib.ptr(flow),
                  _lib.ptr(ws), ws.numel(), _lib.stream(dev), _lib.opt())
    _Staging.done(key2)
    args.flow_result = flow
    pend = Pending(small, asynchronous, 1)
    yield pend
    h = pend.get()
    P = int(h[2 * S])
    if P < 0:
        raise RuntimeError("icpflow_hist_icp abandoned the batch: a wait between workgroups timed out -- a team sharing one "
                           "large pair, or a helper's hand-off in a persistent launch (GPU shared with another process?); "
                           "retry, or register with _lib.options(no_teams=True, no_helpers=True, no_persistent=True)")
    if left_out.any():
        # a pair that was left out of the superset (over-long cluster) and whose clusters both found no partner in stage 1 is
        # a stage-2 candidate of the reference: this path cannot serve it
        b1 = h[:S]
        m_s = b1 >= 0
        m_d = np.zeros(D, dtype=bool)
        m_d[di1[b1[m_s]]] = True
        ls, ld = np.nonzero(left_out)
        if (~m_s[ls] & ~m_d[ld]).any():
            if g is not None:
                g.set_state(gen_state)
            else:
                torch.set_rng_state(gen_state)
            return None
    return rows[:P], T[:P]


def drive(gen):
    """Run a generator of `Pending`s to its end, blocking at every hand-over; -> its return value."""
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


def match_pcds(args, src_points, dst_points, src_labels, dst_labels):
    """utils_match.py:24-66: stage 1 registers clusters that keep their label across the two
    frames (static / slow objects), stage 2 every remaining source cluster against every remaining
    destination cluster.  -> pairs [P,10] (labels, errors, inliers, ratios, ious), transforms [P,4,4].
    Where it can, through ONE call into the library (icpflow_track_frame: the host half in C++ as well, frame_pairs.
    track_frame_native; the random subsamples continue `args.generator` or torch's global generator exactly as torch.randperm
    would) -- same bits as the generators below; `args.native_host = False` / `args.device_association = False` keep to those."""
    if (getattr(args, "native_host", True) and getattr(args, "device_association", None) is not False
            and isinstance(src_points, torch.Tensor) and src_points.is_cuda):
        from . import frame_pairs
        out = frame_pairs.track_frame_native(args, src_points, dst_points, src_labels, dst_labels,
                                             generator=getattr(args, "generator", None) or "global")
        if frame_pairs._served(out):
            return out["pairs"], out["transformations"]
        if out is frame_pairs.NEEDS_HOST_ASSOCIATION:
            from types import SimpleNamespace
            args = SimpleNamespace(**vars(args))
            args.device_association = False     # (the Python host's device path would give up on this frame pair as well)
    return drive(match_pcds_steps(args, src_points, dst_points, src_labels, dst_labels))
