"""CPU restatement of the reference's `cluster_dbscan` branch -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The product package (icp_flow_amd/) must never do so.

What is restated
  * `dbscan_index_order`  -- open3d 0.17.0 `PointCloud::ClusterDBSCAN(eps, min_points)`
    (cpp/open3d/geometry/PointCloudCluster.cpp; pinned by the reference's environment.yml:227 and
    called from utils_cluster.py:34-38).  THIRD-PARTY code that is NOT in /root/reference and not
    installable here; restated from its published algorithm: radius neighbours of every point through
    nanoflann (a neighbour has squared distance STRICTLY below eps^2, the point itself included,
    float32 coordinates widened to double by Vector3dVector), labels start undefined, points are
    visited in index order, a point with fewer than min_points neighbours becomes noise (-1, and may
    later be claimed), any other undefined point seeds cluster `cluster_label` which is grown over
    the neighbours of its core points; noise points reached by the growth take the label.
  * `dbscan_components`   -- the same outcome stated without a visiting order (components of the
    core points, ranked by smallest member; a non-core point joins the lowest-ranked adjacent
    cluster), vectorised for frame-sized inputs.  tests/ check both agree.
  * `keep_largest`        -- the reference's own post-processing, utils_cluster.py:39-46.
  * `cluster_pcd`         -- utils_cluster.py:50-63 (DBSCAN branch).

Pinning: the reference's `cluster_dbscan` / `cluster_pcd` were run in the build container with a
stand-in for open3d backed by sklearn.cluster.DBSCAN (tools/standins/open3d, tools/gen_golden.py
g10); fixtures tests/golden/g10_dbscan.npz.  sklearn keeps a neighbour at distance <= eps, nanoflann
only below eps, so the stand-in searches with the largest double below eps; the generator asserts that
no pair of a fixture falls between the two radii (the demo frame holds 71 pairs at exactly 0.25 m, so
the strict test is exercised).  Open3D itself was never executed: the restatement of its routine is
"parity unpinned" against the library proper.
"""
from collections import deque

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree


def radius_pairs(points, eps):
    """All i < j with squared distance (fp64 on the given coordinates) strictly below eps^2."""
    P = np.ascontiguousarray(points[:, :3], dtype=np.float64)
    if len(P) == 0:
        return np.zeros((0, 2), np.int64)
    cand = cKDTree(P).query_pairs(float(eps) * (1.0 + 1e-9), output_type="ndarray").astype(np.int64)
    if len(cand) == 0:
        return cand.reshape(0, 2)
    d = P[cand[:, 0]] - P[cand[:, 1]]
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    return cand[d2 < float(eps) * float(eps)]


def _csr(n, pairs):
    both = np.concatenate([pairs, pairs[:, ::-1]], axis=0)
    order = np.argsort(both[:, 0], kind="stable")
    both = both[order]
    start = np.searchsorted(both[:, 0], np.arange(n + 1))
    return start, both[:, 1]


def dbscan_index_order(points, eps, min_points):
    """Literal walk of Open3D's loop (pure-Python growth: small and medium inputs)."""
    n = len(points)
    start, nbr = _csr(n, radius_pairs(points, eps))
    degree = np.diff(start) + 1                      # the point itself is in its radius result
    labels = np.full(n, -2, np.int64)
    cluster = 0
    for idx in range(n):
        if labels[idx] != -2:
            continue
        if degree[idx] < min_points:
            labels[idx] = -1
            continue
        labels[idx] = cluster
        todo = deque(nbr[start[idx]:start[idx + 1]].tolist())
        while todo:
            nb = todo.popleft()
            if labels[nb] == -1:
                labels[nb] = cluster
            if labels[nb] != -2:
                continue
            labels[nb] = cluster
            if degree[nb] >= min_points:
                todo.extend(nbr[start[nb]:start[nb + 1]].tolist())
        cluster += 1
    return labels


def dbscan_components(points, eps, min_points):
    n = len(points)
    pairs = radius_pairs(points, eps)
    degree = np.bincount(pairs.reshape(-1), minlength=n) + 1
    core = degree >= min_points
    labels = np.full(n, -1, np.int64)
    if not core.any():
        return labels
    cc = pairs[core[pairs[:, 0]] & core[pairs[:, 1]]]
    _, comp = connected_components(coo_matrix((np.ones(len(cc), np.int8), (cc[:, 0], cc[:, 1])), shape=(n, n)),
                                   directed=False)
    core_idx = np.flatnonzero(core)
    first = np.full(comp.max() + 1, n, np.int64)     # smallest core member of every component
    np.minimum.at(first, comp[core_idx], core_idx)
    live = np.flatnonzero(first < n)
    rank = np.full(comp.max() + 1, -1, np.int64)
    rank[live[np.argsort(first[live])]] = np.arange(len(live))
    labels[core_idx] = rank[comp[core_idx]]
    # non-core points: lowest cluster id among their core neighbours
    best = np.full(n, np.iinfo(np.int64).max)
    for a, b in ((0, 1), (1, 0)):
        sel = ~core[pairs[:, a]] & core[pairs[:, b]]
        np.minimum.at(best, pairs[sel, a], labels[pairs[sel, b]])
    border = ~core & (best < np.iinfo(np.int64).max)
    labels[border] = best[border]
    return labels


def keep_largest(labels, num_clusters):
    """utils_cluster.py:39-46, including its quirk: the first unique label is dropped unseen (it is
    the noise label -1 whenever any point is noise)."""
    labels = labels.copy()
    lbls, counts = np.unique(labels, return_counts=True)
    cluster_info = np.array(list(zip(lbls[1:], counts[1:])))
    cluster_info = cluster_info[cluster_info[:, 1].argsort()]
    clusters_labels = cluster_info[::-1][:num_clusters, 0]
    labels[np.isin(labels, clusters_labels, invert=True)] = -1
    return labels


def cluster_dbscan(args, points, impl=dbscan_components):
    return keep_largest(impl(points, args.epsilon, args.min_cluster_size), args.num_clusters)


def cluster_pcd(args, points, idxs_nonground, impl=dbscan_components):
    """utils_cluster.py:50-63 with if_hdbscan False."""
    labels = np.zeros(len(points)) - 1e8
    labels[idxs_nonground] = cluster_dbscan(args, points[idxs_nonground], impl)
    return labels
