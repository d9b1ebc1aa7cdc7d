"""Drop-ins for the reference's utils_check.py, evaluated for all candidate pairs at once.

The reference loops over candidate pairs in Python and reads device scalars one by one
(utils_check.py:21-49, implicit syncs).  Here the per-cluster statistics are reduced once per cloud
on the device (`ClusterTable`, one workgroup per cluster: `icpflow_cluster_stats`) and come to the
host in ONE transfer (a few hundred clusters x 9 numbers); candidate lists and the three tests are
then a handful of vectorised numpy comparisons, without further device round trips.
"""
import numpy as np
import torch

from . import _lib


def _np(x, dtype=None):
    a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    return a if dtype is None else a.astype(dtype, copy=False)


TABLE_ROWS = 512       # distinct labels a device table holds (the reference keeps num_clusters = 200 + noise + ground)


class ClusterTable:
    """Per-label statistics of one labelled cloud (`icpflow_cluster_table`: one chain of launches).

    device:  points [M,3] float32, order [M] int64 (stable sort of the rows by label: rows of a cluster in original
             order); after `fetch`: labels_unq / count / start [L], mean / extent [L,3] (views of the table)
    host:    h_labels (float32), h_count, h_start (int64), h_mean, h_extent (float32) -- the same numbers
    mean     centroid (utils_check.py:34-35);  extent: sorted bbox extents (get_bbox_tensor,
             utils_helper.py:166-170); both zero for negative labels (ground, noise: never candidates, :32)
    """

    def __init__(self, points, labels, fetch=True, _buffer=None, _launch=True):
        _lib.require_gpu(points, labels)
        self.points = points[:, 0:3].contiguous().float()
        self.labels = labels
        dev = labels.device
        lab = labels.contiguous().float()
        M = int(lab.shape[0])
        self.order = torch.empty((M,), dtype=torch.int64, device=dev)
        # one buffer: [0] holds the int32 number of clusters (first four bytes), [1:] the float64 table [TABLE_ROWS, 9]
        self._packed = _buffer if _buffer is not None else torch.empty((1 + TABLE_ROWS * 9,), dtype=torch.float64, device=dev)
        self._lab = lab
        if _launch:
            ws = _lib.workspace(dev, int(_lib._L.icpflow_cluster_table_workspace_bytes(M, TABLE_ROWS)))
            _lib.call("icpflow_cluster_table", _lib.ptr(self.points), _lib.ptr(lab), M, _lib.ptr(self.order),
                      self._packed.data_ptr() + 8, TABLE_ROWS, _lib.ptr(self._packed), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
        if fetch:
            self.fetch()

    def fetch(self, packed=None):
        """Bring the table to the host: one device -> host transfer (float64 holds the float32 values
        and the counts exactly).  `packed`: this table's part of a transfer shared with other tables."""
        packed = self._packed.cpu().numpy() if packed is None else np.asarray(packed).reshape(-1)
        L = int(packed[0:1].view(np.int32)[0])
        if L < 0:
            self._from_torch()          # more distinct labels than the device table holds: the chain of torch ops
            return
        rows = packed[1: 1 + L * 9].reshape(L, 9)
        self._set_host(rows)
        self._table = self._packed[1: 1 + L * 9].view(L, 9)

    def _set_host(self, rows):
        self.h_labels = rows[:, 0].astype(np.float32)
        self.h_count = rows[:, 1].astype(np.int64)
        self.h_start = rows[:, 2].astype(np.int64)
        self.h_mean = rows[:, 3:6].astype(np.float32)
        self.h_extent = rows[:, 6:9].astype(np.float32)

    def _from_torch(self):
        labels, dev = self._lab, self._lab.device
        self.order = torch.argsort(labels, stable=True)
        labels_unq, count = torch.unique_consecutive(labels[self.order], return_counts=True)
        start = torch.cumsum(count, 0) - count
        L = len(labels_unq)
        labels_unq = labels_unq.float().contiguous()
        mean = torch.empty((L, 3), dtype=torch.float32, device=dev)
        extent = torch.empty((L, 3), dtype=torch.float32, device=dev)
        _lib.call("icpflow_cluster_stats", _lib.ptr(self.points), _lib.ptr(self.order), _lib.ptr(start), _lib.ptr(count),
                  _lib.ptr(labels_unq), L, _lib.ptr(mean), _lib.ptr(extent), _lib.stream(dev))
        self._table = torch.cat([labels_unq.double()[:, None], count.double()[:, None], start.double()[:, None],
                                 mean.double(), extent.double()], dim=1)
        self._set_host(self._table.cpu().numpy())

    # device views of the table (after fetch)
    labels_unq = property(lambda self: self._table[:, 0].float())
    count = property(lambda self: self._table[:, 1].long())
    start = property(lambda self: self._table[:, 2].long())
    mean = property(lambda self: self._table[:, 3:6].float())
    extent = property(lambda self: self._table[:, 6:9].float())

    @staticmethod
    def pair(src_points, src_labels, dst_points, dst_labels, fetch=True):
        """Both tables of a frame pair from ONE chain of launches (icpflow_cluster_table_pair) into ONE buffer: one
        device -> host transfer for the two."""
        both = torch.empty((2, 1 + TABLE_ROWS * 9), dtype=torch.float64, device=src_labels.device)
        st = ClusterTable(src_points, src_labels, fetch=False, _buffer=both[0], _launch=False)
        dt = ClusterTable(dst_points, dst_labels, fetch=False, _buffer=both[1], _launch=False)
        MA, MB = int(st._lab.shape[0]), int(dt._lab.shape[0])
        dev = src_labels.device
        ws = _lib.workspace(dev, int(_lib._L.icpflow_cluster_table_pair_workspace_bytes(MA, MB, TABLE_ROWS)))
        _lib.call("icpflow_cluster_table_pair", _lib.ptr(st.points), _lib.ptr(st._lab), MA, _lib.ptr(st.order),
                  both[0].data_ptr() + 8, _lib.ptr(both[0]), _lib.ptr(dt.points), _lib.ptr(dt._lab), MB, _lib.ptr(dt.order),
                  both[1].data_ptr() + 8, _lib.ptr(both[1]), TABLE_ROWS, _lib.ptr(ws), ws.numel(), _lib.stream(dev))
        st._both = both
        if fetch:
            host = both.cpu().numpy()
            st.fetch(host[0])
            dt.fetch(host[1])
        return st, dt

    def find(self, wanted):
        """Index of each wanted label in labels_unq, or -1 where the cloud has no such cluster
        (device tensors)."""
        unq = self.labels_unq
        pos = torch.searchsorted(unq, wanted.to(unq.dtype))
        pos = pos.clamp(max=len(unq) - 1)
        hit = unq[pos] == wanted.to(unq.dtype)
        return torch.where(hit, pos, torch.full_like(pos, -1))

    def find_host(self, wanted):
        """The same on the host copy: numpy in, numpy out."""
        w = np.asarray(wanted, dtype=np.float32)
        pos = np.minimum(np.searchsorted(self.h_labels, w), len(self.h_labels) - 1)
        return np.where(self.h_labels[pos] == w, pos, -1)


def _sanity_mask(args, st, dt, pairs):
    si, di = st.find_host(pairs[:, 0]), dt.find_host(pairs[:, 1])
    ok = (si >= 0) & (di >= 0)
    s, d = np.maximum(si, 0), np.maximum(di, 0)
    ok &= np.minimum(st.h_count[s], dt.h_count[d]) >= args.min_cluster_size                   # :31
    ok &= np.minimum(pairs[:, 0], pairs[:, 1]) >= 0                                            # :32
    dxy = (dt.h_mean[d] - st.h_mean[s])[:, 0:2]
    ok &= ~(np.sqrt(dxy[:, 0] * dxy[:, 0] + dxy[:, 1] * dxy[:, 1]) > np.float32(args.translation_frame))   # :36
    es, ed = st.h_extent[s], dt.h_extent[d]
    ok &= ~(np.minimum(es, ed) < np.float32(args.thres_box) * np.maximum(es, ed)).any(axis=1)  # :41-43
    return ok


def sanity_grid(args, st, dt, si, di):
    """The same test for every combination of source clusters `si` and destination clusters `di` (rows of
    the tables): -> bool [len(si), len(di)].  Stage 2 of match_pcds tests all remaining sources against all
    remaining destinations (utils_match.py:45-53); on the grid the per-cluster numbers broadcast instead of
    being gathered for each of the S x D candidate rows."""
    # (per-cluster tests first -- size and label are properties of ONE cluster, :31-32 --, the pairwise ones only on
    # the rows and columns that survive them, axis by axis without [S, D, 3] temporaries: this runs on the host thread
    # that also feeds the GPU)
    s_ok = (st.h_count[si] >= args.min_cluster_size) & (st.h_labels[si] >= 0)
    d_ok = (dt.h_count[di] >= args.min_cluster_size) & (dt.h_labels[di] >= 0)
    ok = np.zeros((len(si), len(di)), dtype=bool)
    rs, cs = np.nonzero(s_ok)[0], np.nonzero(d_ok)[0]
    if len(rs) == 0 or len(cs) == 0:
        return ok
    ms, md = st.h_mean[si][rs], dt.h_mean[di][cs]
    dx = md[None, :, 0] - ms[:, None, 0]
    dy = md[None, :, 1] - ms[:, None, 1]
    near = ~(np.sqrt(dx * dx + dy * dy) > np.float32(args.translation_frame))                          # :36
    # (the distance test leaves a few hundred of the S x D combinations: the box test runs on those pairs only)
    pr, pc = np.nonzero(near)
    es, ed = st.h_extent[si][rs][pr], dt.h_extent[di][cs][pc]
    tb = np.float32(args.thres_box)
    keep = ~(np.minimum(es, ed) < tb * np.maximum(es, ed)).any(axis=1)                                  # :41-43
    ok[rs[pr[keep]], cs[pc[keep]]] = True
    return ok


def sanity_check(args, src_table, dst_table, pairs):
    """utils_check.py:21-49 for all candidate `pairs` [K,2] at once -> the surviving rows, in order.
    A pair survives iff both clusters exist with >= min_cluster_size points, both labels are >= 0,
    the xy distance of the centroids is <= translation_frame and, axis by sorted axis, the smaller
    bbox extent is >= thres_box times the larger one.  numpy in -> numpy out, tensor in -> tensor out."""
    if len(pairs) == 0:
        return pairs.reshape(0, 2)
    ok = _sanity_mask(args, src_table, dst_table, _np(pairs, np.float32))
    return pairs[torch.from_numpy(ok).to(pairs.device)] if isinstance(pairs, torch.Tensor) else pairs[ok]


def check_transformation(args, translations, rotations, ious_min):
    """utils_check.py:51-66, vectorised: -> bool [B] (True = keep the match); numpy or tensors."""
    if isinstance(translations, torch.Tensor):
        ok = ~(torch.linalg.norm(translations, dim=1) > args.translation_frame)               # :54
        ok &= ~(ious_min < args.thres_iou)                                                     # :58
        ok &= ~(rotations[:, 1:3].abs().max(dim=1)[0] > args.thres_rot * 90.0)                 # :62-64
        return ok
    t = np.asarray(translations, np.float32)
    ok = ~(np.sqrt((t * t).sum(axis=1, dtype=np.float32)) > np.float32(args.translation_frame))
    ok &= ~(np.asarray(ious_min, np.float32) < np.float32(args.thres_iou))
    ok &= ~(np.abs(np.asarray(rotations, np.float32)[:, 1:3]).max(axis=1) > np.float32(args.thres_rot * 90.0))
    return ok
