"""Stand-in for hdbscan 0.8.29 on top of sklearn.cluster.HDBSCAN (labels differ
from the pinned library, hence cluster labels are a committed fixture)."""
from sklearn.cluster import HDBSCAN as _SkHDBSCAN


class HDBSCAN:
    def __init__(self, min_cluster_size=5, min_samples=None, cluster_selection_epsilon=0.0,
                 alpha=1.0, leaf_size=100, metric="euclidean", **_ignored):
        self._impl = _SkHDBSCAN(min_cluster_size=min_cluster_size, min_samples=min_samples,
                                cluster_selection_epsilon=cluster_selection_epsilon,
                                alpha=alpha, leaf_size=leaf_size, metric=metric)
        self.labels_ = None

    def fit(self, X):
        self._impl.fit(X)
        self.labels_ = self._impl.labels_
        return self
