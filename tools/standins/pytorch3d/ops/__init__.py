from collections import namedtuple

from oracle import core as _core

from . import utils  # noqa: F401

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1,
               return_nn=False, return_sorted=True):
    """Brute-force K=1 nearest neighbour (oracle_core.c:oracle_knn1)."""
    assert K == 1 and norm == 2
    d2, idx, nn = _core.knn1(p1, p2, lengths1, lengths2, return_nn)
    return _KNN(d2[:, :, None], idx[:, :, None], nn[:, :, None, :] if return_nn else None)
