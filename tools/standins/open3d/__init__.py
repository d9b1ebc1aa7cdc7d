"""Stand-in for open3d 0.17.0 (tools/gen_golden.py only; never on the product path).

Import-only for every reference module except utils_cluster.cluster_dbscan, which needs
`o3d.utility.random.seed`, `o3d.utility.Vector3dVector`, `o3d.geometry.PointCloud().points` and
`.cluster_dbscan(eps, min_points)`.  The clustering is backed by sklearn.cluster.DBSCAN on the
float64 copy of the points (Vector3dVector widens to double): same core-point rule (neighbours
within eps, the point itself counted), clusters numbered in order of their first core point, a
non-core point keeps the first cluster that reached it.  sklearn keeps a neighbour at distance <= eps,
Open3D's nanoflann search keeps squared distance < eps^2 only, so the stand-in searches with the
largest double below eps (the generator checks that no pair of a fixture falls between the two radii;
the demo frame holds 71 pairs at exactly 0.25 m).
"""
import types

import numpy as np


class _PointCloud:
    def __init__(self):
        self.points = None

    def cluster_dbscan(self, eps, min_points, print_progress=False):
        from sklearn.cluster import DBSCAN
        pts = np.asarray(self.points, dtype=np.float64)
        return DBSCAN(eps=np.nextafter(float(eps), 0.0), min_samples=min_points, algorithm="kd_tree").fit(pts).labels_.tolist()


geometry = types.SimpleNamespace(PointCloud=_PointCloud)
utility = types.SimpleNamespace(
    Vector3dVector=lambda a: np.asarray(a, dtype=np.float64),
    random=types.SimpleNamespace(seed=lambda s: None),
)
