"""Stand-in for hdbscan 0.8.29 on top of sklearn.cluster.HDBSCAN (labels differ
from the pinned library in their tie-breaking and because the library's default spanning tree is
approximate, hence cluster labels are a committed fixture).

Core-distance convention: hdbscan 0.8.29 on low-dimensional euclidean data (algorithm='best' ->
boruvka_kdtree, hdbscan_.py) queries k = min_samples + 1 neighbours and takes column [min_samples]
(_hdbscan_boruvka.pyx, KDTreeBoruvkaAlgorithm._compute_bounds): the point itself is NOT counted.
sklearn's port counts it ("min_samples ... includes the point itself", sklearn docs).  The stand-in
therefore asks sklearn for min_samples + 1.  [recalled from the library's source; hdbscan is not
installable here]"""
from sklearn.cluster import HDBSCAN as _SkHDBSCAN


class HDBSCAN:
    def __init__(self, min_cluster_size=5, min_samples=None, cluster_selection_epsilon=0.0,
                 alpha=1.0, leaf_size=100, metric="euclidean", **_ignored):
        k = min_cluster_size if min_samples is None else min_samples
        self._impl = _SkHDBSCAN(min_cluster_size=min_cluster_size, min_samples=k + 1,
                                cluster_selection_epsilon=cluster_selection_epsilon,
                                alpha=alpha, leaf_size=leaf_size, metric=metric)
        self.labels_ = None

    def fit(self, X):
        self._impl.fit(X)
        self.labels_ = self._impl.labels_
        return self
