"""The host half of an association stage (utils_match._finish_pairs): reject test, per-source arg-min, threshold -- against the
reference's own formulation with S x D matrices (utils_match.py:96-115, utils_helper.py:108-110) on random stages, ties
included.  Runs without a GPU (the function only reads the stage's results once they are on the host)."""
from types import SimpleNamespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _matrix_form(args, st, dt, si, di, errors, inliers, ratios, ious, translations, rotations, T):
    from icp_flow_amd.utils_check import check_transformation
    keep = check_transformation(args, translations, rotations, np.minimum(ious[:, 0], ious[:, 1]))
    S, D = len(st.h_labels), len(dt.h_labels)
    m_err = np.full((S, D, 2), 1e8, np.float32)
    m_idx = np.full((S, D), -1, np.int64)
    ks = np.nonzero(keep)[0]
    m_err[si[ks], di[ks]] = errors[ks]
    m_idx[si[ks], di[ks]] = ks
    err_min = np.minimum(m_err[:, :, 0], m_err[:, :, 1])
    rows = np.arange(S)
    best = np.argmin(err_min, axis=1)
    valid = err_min[rows, best] < np.float32(args.thres_error)
    rows, best = rows[valid], best[valid]
    k = m_idx[rows, best]
    out = np.concatenate([st.h_labels[rows][:, None], dt.h_labels[best][:, None], errors[k], inliers[k], ratios[k], ious[k]],
                         axis=1).astype(np.float32)
    return out, T[k]


def test_finish_pairs_equals_the_matrix_form_on_random_stages():
    try:
        from icp_flow_amd import utils_match
    except RuntimeError as e:                      # the package refuses to import without its library
        pytest.skip(str(e))
    rng = np.random.default_rng(7)
    args = SimpleNamespace(translation_frame=2.0, thres_iou=0.2, thres_rot=0.1, thres_error=0.2)
    for trial in range(300):
        S, D = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        st = SimpleNamespace(h_labels=np.sort(rng.choice(200, S, replace=False)).astype(np.float32))
        dt = SimpleNamespace(h_labels=np.sort(rng.choice(200, D, replace=False)).astype(np.float32))
        cells = rng.permutation(S * D)[: int(rng.integers(1, S * D + 1))]
        si, di = cells // D, cells % D
        K = len(cells)
        errors = rng.choice([0.03, 0.05, 0.1, 0.15, 0.19, 0.2, 0.3], size=(K, 2)).astype(np.float32)     # many ties
        if trial % 3 == 0:
            errors[rng.random((K, 2)) < 0.1] = np.nan       # (a pair without inliers: np.argmin takes the NaN, the row has no match)
        inliers = rng.integers(0, 500, (K, 2)).astype(np.float32)
        ratios = rng.random((K, 2)).astype(np.float32)
        ious = rng.choice([0.1, 0.2, 0.5, 0.9], size=(K, 2)).astype(np.float32)
        translations = (rng.normal(0, 1.0, (K, 3))).astype(np.float32)
        rotations = (rng.normal(0, 5.0, (K, 3))).astype(np.float32)
        T = rng.normal(0, 1, (K, 4, 4)).astype(np.float32)
        r = np.concatenate([T.reshape(-1), errors.reshape(-1), inliers.reshape(-1), ratios.reshape(-1), ious.reshape(-1),
                            translations.reshape(-1), rotations.reshape(-1), np.array([7], np.int32).view(np.float32)])
        got_rows, got_T = utils_match._finish_pairs(args, st, dt, (si, di, r))
        want_rows, want_T = _matrix_form(args, st, dt, si, di, errors, inliers, ratios, ious, translations, rotations, T)
        assert np.array_equal(got_rows, want_rows), trial
        assert np.array_equal(got_T, want_T), trial


def test_sanity_grid_equals_the_row_wise_test_on_random_tables():
    """utils_check.sanity_grid (stage 2: every remaining source against every remaining destination, on the S x D grid) against
    utils_check._sanity_mask row by row (utils_check.py:21-49) on random cluster tables: sizes around min_cluster_size, centroids
    around translation_frame apart, box extents around the thres_box ratio, negative labels."""
    try:
        from icp_flow_amd import utils_check
    except RuntimeError as e:
        pytest.skip(str(e))

    class Table(SimpleNamespace):
        find_host = utils_check.ClusterTable.find_host

    rng = np.random.default_rng(11)
    args = SimpleNamespace(min_cluster_size=20, translation_frame=2.0, thres_box=0.1)
    for trial in range(60):
        tabs = []
        for n in (int(rng.integers(1, 40)), int(rng.integers(1, 40))):
            lab = np.sort(rng.choice(np.arange(-1, 90), n, replace=False)).astype(np.float64)
            rows = np.stack([lab, rng.integers(5, 60, n).astype(np.float64), np.zeros(n), *(rng.uniform(-3, 3, (3, n))),
                             *np.sort(rng.choice([0.05, 0.1, 0.5, 1.0, 4.0], (n, 3)), axis=1).T], axis=1)
            t = Table()
            utils_check.ClusterTable._set_host(t, rows)
            tabs.append(t)
        st, dt = tabs
        si, di = np.arange(len(st.h_labels)), np.arange(len(dt.h_labels))
        got = utils_check.sanity_grid(args, st, dt, si, di)
        pairs = np.stack(np.meshgrid(st.h_labels, dt.h_labels, indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
        want = utils_check._sanity_mask(args, st, dt, pairs).reshape(len(si), len(di))
        assert np.array_equal(got, want), trial
