"""Drop-ins for the reference's utils_check.py, evaluated for all candidate pairs at once.

The reference loops over candidate pairs in Python and reads device scalars one by one
(utils_check.py:21-49, implicit syncs); here per-cluster statistics are reduced once per cloud
(`ClusterTable`, one workgroup per cluster: `icpflow_cluster_stats`) and the three tests are a handful of vectorised comparisons.
"""
import torch

from . import _lib


class ClusterTable:
    """Per-label statistics of one labelled cloud, all on the cloud's device.

    labels_unq  [L]    sorted unique labels (float, like the reference: ground -1e8, noise -1)
    count       [L]    points per label
    start       [L]    first row of the label in `order`
    order       [M]    stable argsort of the labels: rows of a cluster in original order
    mean        [L,3]  centroid (fp32 mean, utils_check.py:34-35)
    extent      [L,3]  sorted axis-aligned bbox extents (get_bbox_tensor, utils_helper.py:166-170)
    """

    def __init__(self, points, labels):
        self.points = points[:, 0:3].contiguous().float()
        self.labels = labels
        self.order = torch.argsort(labels, stable=True)
        sorted_labels = labels[self.order]
        self.labels_unq, self.count = torch.unique_consecutive(sorted_labels, return_counts=True)
        self.start = torch.cumsum(self.count, 0) - self.count
        L = len(self.labels_unq)
        self.mean = torch.empty((L, 3), dtype=torch.float32, device=labels.device)
        self.extent = torch.empty((L, 3), dtype=torch.float32, device=labels.device)
        _lib.require_gpu(self.points, labels)
        _lib.call("icpflow_cluster_stats", _lib.ptr(self.points), _lib.ptr(self.order), _lib.ptr(self.start),
                  _lib.ptr(self.count), L, _lib.ptr(self.mean), _lib.ptr(self.extent), _lib.stream(labels.device))

    def find(self, wanted):
        """Index of each wanted label in labels_unq, or -1 where the cloud has no such cluster."""
        pos = torch.searchsorted(self.labels_unq, wanted.to(self.labels_unq.dtype))
        pos = pos.clamp(max=len(self.labels_unq) - 1)
        hit = self.labels_unq[pos] == wanted.to(self.labels_unq.dtype)
        return torch.where(hit, pos, torch.full_like(pos, -1))


def sanity_check(args, src_table, dst_table, pairs):
    """utils_check.py:21-49 for all candidate `pairs` [K,2] at once -> the surviving rows, in order.
    A pair survives iff both clusters exist with >= min_cluster_size points, both labels are >= 0,
    the xy distance of the centroids is <= translation_frame and, axis by sorted axis, the smaller
    bbox extent is >= thres_box times the larger one."""
    if len(pairs) == 0:
        return pairs.reshape(0, 2)
    si = src_table.find(pairs[:, 0])
    di = dst_table.find(pairs[:, 1])
    ok = (si >= 0) & (di >= 0)
    s = si.clamp(min=0)
    d = di.clamp(min=0)
    ok &= torch.minimum(src_table.count[s], dst_table.count[d]) >= args.min_cluster_size   # :31
    ok &= pairs.min(dim=1)[0] >= 0                                                           # :32
    dxy = (dst_table.mean[d] - src_table.mean[s])[:, 0:2]
    ok &= ~(torch.linalg.norm(dxy, dim=1) > args.translation_frame)                          # :36
    es, ed = src_table.extent[s], dst_table.extent[d]
    ok &= ~(torch.minimum(es, ed) < args.thres_box * torch.maximum(es, ed)).any(dim=1)       # :41-43
    return pairs[ok]


def check_transformation(args, translations, rotations, ious_min):
    """utils_check.py:51-66, vectorised: -> bool [B] (True = keep the match)."""
    ok = ~(torch.linalg.norm(translations, dim=1) > args.translation_frame)                  # :54
    ok &= ~(ious_min < args.thres_iou)                                                       # :58
    ok &= ~(rotations[:, 1:3].abs().max(dim=1)[0] > args.thres_rot * 90.0)                   # :62-64
    return ok
