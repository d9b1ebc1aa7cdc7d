"""Drop-in for the reference's utils_cluster (utils_cluster.py:32-63), DBSCAN branch, on the GPU.

`cluster_dbscan(args, points)` and `cluster_pcd(args, points, idxs_nonground)` keep the reference's
names, arguments and label conventions: cluster ids in Open3D's numbering (order of each cluster's first
core point), -1 for unclustered points, -1e8 for ground rows; only the `args.num_clusters` largest
clusters survive (utils_cluster.py:39-46, quirks included).  numpy in -> numpy out like the reference;
a GPU tensor in -> GPU tensor out (labels stay resident for ClusterTable / match_pcds).

The neighbour search, core test, component labelling and sizes run in libicpflow_hip.so
(`icpflow_dbscan`, csrc/cluster.hip); the choice of the clusters to keep is the reference's own numpy
expression applied to the C cluster sizes (a few hundred numbers).  `cluster_hdbscan` is NOT built:
the hdbscan package's approximate Boruvka spanning tree is not reproducible here.
"""
import numpy as np
import torch

from . import _lib


def _device_points(points):
    if isinstance(points, torch.Tensor):
        _lib.require_gpu(points)
        return points, True
    if not torch.cuda.is_available():
        raise RuntimeError("icp_flow_amd.utils_cluster: no GPU (HIP) device -- there is no CPU path")
    return torch.from_numpy(np.ascontiguousarray(points)).cuda(), False


def dbscan(points, eps, min_points, mask=None):
    """-> (labels int32 [n] on the GPU: id / -1 noise / -2 masked out, sizes int32 [C] on the GPU)."""
    pts, _ = _device_points(points)
    if pts.dim() != 2 or pts.shape[1] < 3:
        raise RuntimeError(f"dbscan: expected points [n, >=3], got {tuple(pts.shape)}")
    pts = pts.float().contiguous()
    dev = pts.device
    n = pts.shape[0]
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    if n == 0:
        return labels, torch.empty(0, dtype=torch.int32, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    num = torch.empty(1, dtype=torch.int32, device=dev)
    m = None
    if mask is not None:
        m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask))
        m = m.to(dev).to(torch.uint8).contiguous()
        if m.shape != (n,):
            raise RuntimeError(f"dbscan: mask must have shape ({n},), got {tuple(m.shape)}")
    ws = _lib.workspace(dev, int(_lib._L.icpflow_dbscan_workspace_bytes(n)))
    _lib.call("icpflow_dbscan", _lib.ptr(pts), pts.shape[1], _lib.ptr(m), n, float(eps), int(min_points),
              _lib.ptr(labels), _lib.ptr(counts), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
    return labels, counts[: int(num.item())]


def _kept_clusters(sizes, n_noise, num_clusters):
    """utils_cluster.py:39-45 on (unique labels, counts): the first unique label is skipped unseen --
    it is -1 whenever any point is unclustered, cluster 0 otherwise -- then the clusters are ordered by
    numpy's default argsort of their sizes and the last num_clusters survive."""
    lbls = np.arange(len(sizes), dtype=np.int64)
    counts = np.asarray(sizes, dtype=np.int64)
    if n_noise > 0:
        lbls, counts = np.concatenate([[-1], lbls]), np.concatenate([[n_noise], counts])
    cluster_info = np.array(list(zip(lbls[1:], counts[1:])))
    cluster_info = cluster_info[cluster_info[:, 1].argsort()]   # IndexError when nothing is left, as upstream
    return cluster_info[::-1][:num_clusters, 0]


def _cluster(args, points, mask):
    labels, sizes = dbscan(points, args.epsilon, args.min_cluster_size, mask)
    sizes_h = sizes.cpu().numpy()
    n_live = int(labels.numel() if mask is None else (labels > -2).sum().item())
    keep_ids = _kept_clusters(sizes_h, n_live - int(sizes_h.sum()), args.num_clusters)
    keep = torch.zeros(len(sizes_h) + 1, dtype=torch.bool, device=labels.device)
    keep[torch.from_numpy(np.ascontiguousarray(keep_ids[keep_ids >= 0])).to(labels.device)] = True
    lab = labels.long()
    kept = keep[lab.clamp(min=0)] & (lab >= 0)
    return torch.where(kept, lab, torch.where(lab == -2, lab, torch.full_like(lab, -1)))


def cluster_dbscan(args, points):
    """utils_cluster.py:32-48.  Returns integer labels, -1 = unclustered."""
    _, resident = _device_points(points)
    lab = _cluster(args, points, None)
    return lab if resident else lab.cpu().numpy()


def cluster_pcd(args, points, idxs_nonground):
    """utils_cluster.py:50-63: float64 labels, ground rows -1e8, the rest from cluster_dbscan of the
    non-ground rows (the mask is applied inside the kernels; no compacted copy is made)."""
    if getattr(args, "if_hdbscan", False):
        return cluster_hdbscan(args, points)
    _, resident = _device_points(points)
    if isinstance(idxs_nonground, torch.Tensor):
        mask = idxs_nonground
    else:
        mask = np.asarray(idxs_nonground)
    if mask.dtype not in (torch.bool, np.bool_, np.dtype(bool)):   # index list -> mask
        full = np.zeros(len(points), dtype=bool)
        full[np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)] = True
        mask = full
    lab = _cluster(args, points, mask)
    out = torch.where(lab == -2, torch.full((), -1e8, dtype=torch.float64, device=lab.device), lab.double())
    return out if resident else out.cpu().numpy()


def cluster_hdbscan(args, points):
    raise NotImplementedError(
        "icp_flow_amd: cluster_hdbscan (utils_cluster.py:10-29) is not built -- pass labels computed upstream, "
        "or use the DBSCAN branch (if_hdbscan=False); see DESIGN.md section 8")
