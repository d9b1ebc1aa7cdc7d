"""Drop-in for the reference's utils_cluster (utils_cluster.py:32-63), DBSCAN branch, on the GPU.

`cluster_dbscan(args, points)` and `cluster_pcd(args, points, idxs_nonground)` keep the reference's
names, arguments and label conventions: cluster ids in Open3D's numbering (order of each cluster's first
core point), -1 for unclustered points, -1e8 for ground rows; only the `args.num_clusters` largest
clusters survive (utils_cluster.py:39-46, quirks included).  numpy in -> numpy out like the reference;
a GPU tensor in -> GPU tensor out (labels stay resident for ClusterTable / match_pcds).

The neighbour search, core test, component labelling and sizes run in libicpflow_hip.so
(`icpflow_dbscan`, csrc/cluster.hip); the choice of the clusters to keep is the reference's own numpy
expression applied to the C cluster sizes (a few hundred numbers).

`cluster_hdbscan(args, points)` (utils_cluster.py:10-29): the O(n^2) part -- core distances and the exact
minimum spanning tree of the mutual-reachability graph -- runs in libicpflow_hip.so (`icpflow_hdbscan_mst`,
csrc/hdbscan.hip); the sequential remainder on the n - 1 tree edges (sort, dendrogram, condensed tree,
excess-of-mass selection) is host C++ in the same library (`icpflow_hdbscan_labels`, csrc/hdbscan_tree.cpp),
checked against scikit-learn's compiled routines for those steps.
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
    _, resident = _device_points(points)
    if isinstance(idxs_nonground, torch.Tensor):
        mask = idxs_nonground
    else:
        mask = np.asarray(idxs_nonground)
    if mask.dtype not in (torch.bool, np.bool_, np.dtype(bool)):   # index list -> mask
        full = np.zeros(len(points), dtype=bool)
        full[np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)] = True
        mask = full
    if getattr(args, "if_hdbscan", False):
        lab_h = cluster_hdbscan(args, points, mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask)
        out_h = np.where(lab_h == -2, -1e8, lab_h.astype(np.float64))
        return torch.from_numpy(out_h).to(points.device) if resident else out_h
    lab = _cluster(args, points, mask)
    out = torch.where(lab == -2, torch.full((), -1e8, dtype=torch.float64, device=lab.device), lab.double())
    return out if resident else out.cpu().numpy()


def hdbscan_mst(points, min_samples, mask=None, cell=0.25):
    """Core distances and the minimum spanning tree of the mutual-reachability graph on the GPU.
    -> dict of GPU tensors: a, b (int32 caller rows), w2 (float64 squared weights) of the n_live - 1 tree
    edges, core2 (float64 [n] squared core distance, NaN for rows that took no part), n_live (int)."""
    pts, _ = _device_points(points)
    if pts.dim() != 2 or pts.shape[1] < 3:
        raise RuntimeError(f"hdbscan_mst: expected points [n, >=3], got {tuple(pts.shape)}")
    pts = pts.float().contiguous()
    dev = pts.device
    n = pts.shape[0]
    if n == 0:
        raise RuntimeError("hdbscan_mst: no points")
    m = None
    if mask is not None:
        m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask))
        m = m.to(dev).to(torch.uint8).contiguous()
        if m.shape != (n,):
            raise RuntimeError(f"hdbscan_mst: mask must have shape ({n},), got {tuple(m.shape)}")
    a = torch.empty(n, dtype=torch.int32, device=dev)
    b = torch.empty(n, dtype=torch.int32, device=dev)
    w2 = torch.empty(n, dtype=torch.float64, device=dev)
    core2 = torch.empty(n, dtype=torch.float64, device=dev)
    cnt = torch.empty(2, dtype=torch.int32, device=dev)
    ws = _lib.workspace(dev, int(_lib._L.icpflow_hdbscan_mst_workspace_bytes(n)))
    _lib.call("icpflow_hdbscan_mst", _lib.ptr(pts), pts.shape[1], _lib.ptr(m), n, int(min_samples), float(cell),
              _lib.ptr(core2), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w2), _lib.ptr(cnt[0:1]), _lib.ptr(cnt[1:2]),
              _lib.ptr(ws), ws.numel(), _lib.stream(dev))
    ne, nl = (int(v) for v in cnt.cpu())
    if nl < int(min_samples):
        raise ValueError(f"hdbscan_mst: {nl} points have no {int(min_samples)}-th nearest neighbour (min_samples)")
    if ne != max(nl - 1, 0):
        raise RuntimeError(f"hdbscan_mst: {ne} tree edges for {nl} points (internal error)")
    return dict(a=a[:ne], b=b[:ne], w2=w2[:ne], core2=core2, n_live=nl)


def labels_from_mst(a, b, w, n, min_cluster_size):
    """The sequential remainder of HDBSCAN on the n - 1 tree edges (host C++ in libicpflow_hip.so,
    csrc/hdbscan_tree.cpp): edges directed away from point 0 and sorted by weight, single-linkage dendrogram,
    condensed tree with min_cluster_size, excess-of-mass selection.  -> int64 labels [n], -1 noise."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    if len(a) != n - 1 or len(b) != n - 1 or len(w) != n - 1:
        raise ValueError(f"labels_from_mst: {len(a)} edges cannot span {n} points")
    out = np.empty(n, dtype=np.int32)
    rc = _lib._L.icpflow_hdbscan_labels(a.ctypes.data, b.ctypes.data, w.ctypes.data, int(n), int(min_cluster_size),
                                        out.ctypes.data)
    if rc != 0:
        raise ValueError(f"labels_from_mst: invalid tree or min_cluster_size (code {rc})")
    return out.astype(np.int64)


def hdbscan(points, min_cluster_size, min_samples=None, mask=None, cell=0.25, counts_self=False):
    """hdbscan.HDBSCAN(min_cluster_size, min_samples).fit(points).labels_ for euclidean points, alpha 1, EOM
    selection.  -> int64 numpy labels [n]: cluster id, -1 noise (and non-finite rows), -2 masked out.

    Core distance = distance to the min_samples-th nearest neighbour NOT counting the point itself: the convention
    of the `hdbscan` package on the reference's path (0.8.29, algorithm 'best' -> boruvka_kdtree on 3-D euclidean
    data: _hdbscan_boruvka.pyx queries k = min_samples + 1 and takes column [min_samples]).  counts_self=True
    gives scikit-learn's convention ("includes the point itself"): one neighbour fewer."""
    k = int(min_cluster_size if min_samples is None else min_samples) + (0 if counts_self else 1)
    t = hdbscan_mst(points, k, mask, cell)
    n = int(t["core2"].numel())
    live = torch.isfinite(t["core2"]) | torch.isinf(t["core2"])   # NaN marks rows that took no part
    live_h = live.cpu().numpy()
    nl = t["n_live"]
    if nl < max(k, 2):
        raise ValueError(f"hdbscan: {nl} points cannot be clustered with min_samples {k}")
    # the edges leave the GPU in ascending weight (a device sort costs microseconds, the host's radix passes were a
    # fifth of its share); ties are put in their fixed order by the host code either way
    order = torch.argsort(t["w2"])
    a, b = t["a"][order].cpu().numpy(), t["b"][order].cpu().numpy()
    if nl != n:                                                     # caller row -> row of the clustered subset
        sub = np.cumsum(live_h) - 1
        a, b = sub[a], sub[b]
    lab = labels_from_mst(a, b, np.sqrt(t["w2"][order].cpu().numpy()), nl, min_cluster_size)
    out = np.full(n, -1, dtype=np.int64)
    out[live_h] = lab
    if mask is not None:
        mh = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
        out[~mh.astype(bool)] = -2
    return out


def cluster_hdbscan(args, points, mask=None):
    """utils_cluster.py:10-29: HDBSCAN(min_cluster_size=args.min_cluster_size, min_samples=None), then only the
    args.num_clusters largest clusters survive (same numpy expression as the DBSCAN branch, :19-26)."""
    _, resident = _device_points(points)
    lab = hdbscan(points, args.min_cluster_size, None, mask)
    keep = lab >= -1
    hist = np.bincount(lab[keep] + 1)                               # == np.unique(lab[keep], return_counts=True) without
    lbls = np.flatnonzero(hist) - 1                                  # the sort of 10^5 labels: values, ascending, and
    counts = hist[lbls + 1]                                          # their counts
    cluster_info = np.array(list(zip(lbls[1:], counts[1:])))
    cluster_info = cluster_info[cluster_info[:, 1].argsort()]
    clusters_labels = cluster_info[::-1][:args.num_clusters, 0]
    kept = np.zeros(int(lab.max()) + 2, dtype=bool)                 # label + 1 -> survives (table instead of isin)
    kept[np.asarray(clusters_labels, dtype=np.int64) + 1] = True
    lab[keep & ~kept[np.maximum(lab, -1) + 1]] = -1
    if mask is None:
        return torch.from_numpy(lab).to(points.device) if resident else lab
    return lab
