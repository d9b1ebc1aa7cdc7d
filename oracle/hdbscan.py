"""CPU restatement of the spanning-tree part of HDBSCAN -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The product package (icp_flow_amd/) must never do so.

The reference's `cluster_hdbscan` (utils_cluster.py:10-29) calls the third-party `hdbscan` 0.8.29
(environment.yml:57; not in /root/reference, not installable here).  Its published algorithm (Campello et
al. 2013; McInnes & Healy 2017), as also implemented by scikit-learn's port
(sklearn/cluster/_hdbscan/hdbscan.py:_hdbscan_prims, _linkage.pyx:mst_from_data_matrix):
  core(a)      = distance from a to its k-th nearest neighbour, a itself counted (tree.query(X, k)[0][:, -1]);
                 scikit-learn takes k = min_samples, the hdbscan package k = min_samples + 1 (it does not count
                 the point: _hdbscan_boruvka.pyx).  `min_samples` below is that k, the point counted,
  d_mreach(a,b) = max(core(a), core(b), |a - b| / alpha),  alpha = 1,
  a minimum spanning tree of the complete graph under d_mreach, then the single-linkage dendrogram,
  the condensed tree (min_cluster_size) and the excess-of-mass selection.
Restated here, dense and exact, for small inputs:
  * `core2`          squared core distances (fp64, dx*dx + dy*dy + dz*dz left to right, no contraction),
  * `mst`            THE minimum spanning tree under the strict total order (squared weight, smaller row,
                     larger row) -- the tie rule of the HIP kernels, so trees compare edge for edge.
The pinned library builds an APPROXIMATE tree (approx_min_span_tree=True) and sklearn's Prim breaks ties by
visiting order, so labels are pinned against the reference run only up to points at tied merge heights:
fixtures tests/golden/g11_hdbscan.npz and g8_demo_labels.npz hold the reference's cluster_pcd output with
sklearn's HDBSCAN standing in for the hdbscan package (tools/standins/hdbscan) -- "parity unpinned" against
the hdbscan package proper.
"""
import numpy as np
from scipy.sparse.csgraph import minimum_spanning_tree


def sq_dists(points):
    P = np.ascontiguousarray(points[:, :3], dtype=np.float64)
    d = P[:, None, :] - P[None, :, :]
    return d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1] + d[:, :, 2] * d[:, :, 2]


def core2(points, min_samples):
    D2 = sq_dists(points)
    return np.partition(D2, min_samples - 1, axis=1)[:, min_samples - 1]


def mst(points, min_samples):
    """-> (a, b, w2) sorted by (a, b) with a < b: the unique tree under (w2, a, b) ordering."""
    D2 = sq_dists(points)
    n = len(D2)
    c2 = np.partition(D2, min_samples - 1, axis=1)[:, min_samples - 1]
    W2 = np.maximum(np.maximum(c2[:, None], c2[None, :]), D2)
    iu = np.triu_indices(n, 1)
    w = W2[iu]
    order = np.lexsort((iu[1], iu[0], w))                # by weight, then smaller row, then larger row
    rank = np.empty(len(w), dtype=np.float64)
    rank[order] = np.arange(1, len(w) + 1, dtype=np.float64)
    R = np.zeros((n, n), dtype=np.float64)
    R[iu] = rank                                         # unique positive weights: the tree is unique
    T = minimum_spanning_tree(R).tocoo()
    a, b = np.minimum(T.row, T.col), np.maximum(T.row, T.col)
    o = np.lexsort((b, a))
    a, b = a[o], b[o]
    return a.astype(np.int64), b.astype(np.int64), W2[a, b], c2
