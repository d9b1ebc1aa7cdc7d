"""DBSCAN clustering of a frame pair (utils_cluster.py:32-63, DBSCAN branch).

CPU: the oracle (oracle/cluster.py) against G10 = the reference's cluster_pcd run with the open3d
stand-in (tools/gen_golden.py g10), and the two statements of the algorithm against each other.
GPU: icp_flow_amd.utils_cluster (icpflow_dbscan through the C ABI) against G10 and the oracle,
bit-exact (labels are indices)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cluster as oc


def _args(params):
    eps, mcs, ncl = params
    return SimpleNamespace(epsilon=float(eps), min_cluster_size=int(mcs), num_clusters=int(ncl), if_hdbscan=False)


def _demo_points():
    g = load_golden("g8_demo")
    return np.concatenate([g["point_dst"], g["point_src"]], axis=0)   # demo.py:210


def _cloud(seed, n, spread=8.0, sigma=0.25, k=10):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-spread, spread, size=(k, 3)) * np.array([1, 1, 0.2])
    p = centers[rng.integers(0, k, n)] + rng.normal(0, sigma, size=(n, 3)) * np.array([1, 1, 0.5])
    p[: n // 5] = rng.uniform(-spread - 2, spread + 2, size=(n // 5, 3)) * np.array([1, 1, 0.2])
    return p.astype(np.float32)


# ----------------------------------------------------------------------------- CPU: oracle vs golden
@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_small_clouds_match_reference_run(k):
    g = load_golden("g10_dbscan")
    p, ng, want = g[f"small_{k}_points"], g[f"small_{k}_nonground"], g[f"small_{k}_labels"]
    a = _args(g[f"small_{k}_params"])
    assert np.array_equal(oc.cluster_pcd(a, p, ng), want)
    assert np.array_equal(oc.cluster_pcd(a, p, ng, impl=oc.dbscan_index_order), want)
    assert (want[~ng] == -1e8).all() and (want[ng] >= -1).all()


def test_oracle_demo_frame_matches_reference_run():
    g = load_golden("g10_dbscan")
    pts = _demo_points()
    got = oc.cluster_pcd(_args(g["demo_a_params"]), pts, np.ones(len(pts), dtype=bool))
    assert np.array_equal(got.astype(np.int32), g["demo_a_labels"])
    assert int(g["demo_a_pairs_at_eps"]) == 71      # the strict radius test is exercised by the fixture


def test_oracle_visiting_order_statement_equals_component_statement():
    for seed, n, eps, mp in [(1, 700, 0.3, 5), (2, 1500, 0.2, 3), (3, 400, 0.5, 12), (4, 900, 0.25, 1)]:
        p = _cloud(seed, n)
        assert np.array_equal(oc.dbscan_index_order(p, eps, mp), oc.dbscan_components(p, eps, mp))


def test_oracle_strict_radius_on_a_lattice():
    # points spaced exactly eps apart are NOT neighbours (nanoflann keeps dist < radius)
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(2), indexing="ij"), -1).reshape(-1, 3)
    p = (g * 0.25).astype(np.float32)
    assert (oc.dbscan_components(p, 0.25, 2) == -1).all()
    assert (oc.dbscan_components(p, 0.2500001, 2) == 0).all()


# ----------------------------------------------------------------------------- GPU: product vs both
gpu = pytest.mark.gpu


def _hip():
    from icp_flow_amd import utils_cluster
    return utils_cluster


@gpu
@pytest.mark.parametrize("k", [0, 1, 2])
def test_gpu_small_clouds_bit_exact(k):
    g = load_golden("g10_dbscan")
    p, ng, want = g[f"small_{k}_points"], g[f"small_{k}_nonground"], g[f"small_{k}_labels"]
    got = _hip().cluster_pcd(_args(g[f"small_{k}_params"]), p, ng)
    assert isinstance(got, np.ndarray) and got.dtype == np.float64
    assert np.array_equal(got, want)


@gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gpu_demo_frame_bit_exact(tag):
    g = load_golden("g10_dbscan")
    pts = _demo_points()
    a = _args(g[f"demo_{tag}_params"])
    got = _hip().cluster_pcd(a, pts, np.ones(len(pts), dtype=bool))
    assert np.array_equal(got.astype(np.int32), g[f"demo_{tag}_labels"])
    # resident variant: GPU tensor in, GPU tensor out, same labels; index list instead of a mask
    t = torch.from_numpy(pts).cuda()
    res = _hip().cluster_pcd(a, t, torch.ones(len(pts), dtype=torch.bool, device="cuda"))
    assert res.is_cuda and res.dtype == torch.float64
    assert np.array_equal(res.cpu().numpy().astype(np.int32), g[f"demo_{tag}_labels"])


@gpu
@pytest.mark.parametrize("seed,n,eps,mp", [(11, 1, 0.3, 1), (12, 2, 0.3, 2), (13, 257, 0.3, 4), (14, 5000, 0.2, 6),
                                           (15, 20000, 0.35, 10), (16, 3000, 0.05, 3), (17, 3000, 3.0, 40),
                                           (18, 4096, 0.25, 1)])
def test_gpu_raw_labels_and_sizes_vs_oracle(seed, n, eps, mp):
    p = _cloud(seed, n)
    lab, sizes = _hip().dbscan(p, eps, mp)
    want = oc.dbscan_components(p, eps, mp)
    assert np.array_equal(lab.cpu().numpy(), want)
    c = int(want.max()) + 1
    assert sizes.numel() == c
    assert np.array_equal(sizes.cpu().numpy(), np.bincount(want[want >= 0], minlength=c))


@gpu
def test_gpu_edge_cases():
    hip = _hip()
    # exact-eps lattice: strict test
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(2), indexing="ij"), -1).reshape(-1, 3)
    p = (g * 0.25).astype(np.float32)
    assert (hip.dbscan(p, 0.25, 2)[0].cpu().numpy() == -1).all()
    assert (hip.dbscan(p, 0.2500001, 2)[0].cpu().numpy() == 0).all()
    # duplicates: 50 copies of one point = one cluster; min_points above the copies = noise
    d = np.tile(np.array([[1.0, -2.0, 0.5]], dtype=np.float32), (50, 1))
    assert (hip.dbscan(d, 0.1, 50)[0].cpu().numpy() == 0).all()
    assert (hip.dbscan(d, 0.1, 51)[0].cpu().numpy() == -1).all()
    # everything in one grid cell, far from the origin, wide rows (stride 4 with a flag column)
    rng = np.random.default_rng(5)
    q = np.concatenate([rng.normal(0, 0.02, (800, 3)) + np.array([5000.0, -7000.0, 30.0]), np.ones((800, 1))], 1)
    q = q.astype(np.float32)
    assert np.array_equal(hip.dbscan(q, 0.25, 5)[0].cpu().numpy(), oc.dbscan_components(q, 0.25, 5))
    # non-finite rows never cluster and never join; masked rows report -2 and do not bridge clusters
    p = _cloud(21, 2000)
    bad = p.copy()
    bad[::97, 1] = np.nan
    bad[5::131, 0] = np.inf
    fin = np.isfinite(bad).all(1)
    want = np.full(len(bad), -1, np.int64)
    want[fin] = oc.dbscan_components(bad[fin], 0.3, 5)
    assert np.array_equal(hip.dbscan(bad, 0.3, 5)[0].cpu().numpy(), want)
    mask = rng.random(len(p)) < 0.7
    want = np.full(len(p), -2, np.int64)
    want[mask] = oc.dbscan_components(p[mask], 0.3, 5)
    assert np.array_equal(hip.dbscan(p, 0.3, 5, mask)[0].cpu().numpy(), want)
    # a cloud wider than the 21-bit cell range (cells clamp, adjacency survives)
    far = (_cloud(22, 3000) * np.array([1e5, 1.0, 1.0])).astype(np.float32)
    far[:1500] = _cloud(23, 1500) + np.array([3e6, 0, 0], dtype=np.float32)
    assert np.array_equal(hip.dbscan(far, 0.3, 4)[0].cpu().numpy(), oc.dbscan_components(far, 0.3, 4))


@gpu
def test_gpu_keep_largest_quirks_and_errors():
    hip = _hip()
    p = _cloud(31, 4000)
    for ncl in (1, 3, 1000):
        a = SimpleNamespace(epsilon=0.3, min_cluster_size=5, num_clusters=ncl, if_hdbscan=False)
        assert np.array_equal(hip.cluster_dbscan(a, p), oc.cluster_dbscan(a, p))
    # no unclustered point at all: upstream drops cluster 0 unseen (utils_cluster.py:41)
    two = np.concatenate([np.zeros((30, 3)), np.ones((40, 3)) * 5, np.ones((35, 3)) * 9]).astype(np.float32)
    a = SimpleNamespace(epsilon=0.3, min_cluster_size=5, num_clusters=5, if_hdbscan=False)
    got = hip.cluster_dbscan(a, two)
    assert np.array_equal(got, oc.cluster_dbscan(a, two))
    assert (got[:30] == -1).all() and (got[30:70] == 1).all()
    # nothing clusters: upstream's IndexError
    lone = (np.arange(30)[:, None] * np.array([[10.0, 0, 0]])).astype(np.float32)
    with pytest.raises(IndexError):
        oc.cluster_dbscan(a, lone)
    with pytest.raises(IndexError):
        hip.cluster_dbscan(a, lone)
    with pytest.raises(RuntimeError):
        hip.dbscan(torch.zeros(10, 3), 0.3, 5)             # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        hip.dbscan(p, -1.0, 5)
    with pytest.raises(RuntimeError):
        hip.dbscan(p, 0.3, 0)


@gpu
def test_gpu_ground_threshold_then_cluster_matches_reference_flow():
    from icp_flow_amd import utils_ground
    g = load_golden("g10_dbscan")
    p = g["small_2_points"]
    a = _args(g["small_2_params"])
    a.range_z, a.ground_slack = -0.8, 0.3             # z <= -0.5 is ground: the fixture's mask
    ng = utils_ground.segment_ground_thres(a, p)
    assert np.array_equal(ng, g["small_2_nonground"])
    assert np.array_equal(_hip().cluster_pcd(a, p, ng), g["small_2_labels"])


@gpu
def test_gpu_unlabelled_frame_pair_is_clustered_then_registered(tmp_path):
    """demo frame pair WITHOUT labels through the stream: joint DBSCAN on the GPU (demo.py:210) + track +
    flow equals the same registration fed with the CPU port's labels; flow error against ground truth is
    that of a sensible clustering."""
    from icp_flow_amd import frame_pairs
    g = load_golden("g8_demo")
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], None, None, None, g["gt_flow"])
    frame_pairs.save_frame_pair(str(tmp_path / "demo.npz"), fp)
    back = frame_pairs.load_frame_pair(str(tmp_path / "demo.npz"))
    assert back.labels_src is None
    dev = torch.device("cuda", 0)
    a = frame_pairs.default_args(max_points=2048, cluster="dbscan", epsilon=0.25, min_cluster_size=20, num_clusters=200)
    got = frame_pairs.register_frame_pair(a, back, dev)
    ca = SimpleNamespace(epsilon=0.25, min_cluster_size=20, num_clusters=200, if_hdbscan=False)
    lab = oc.cluster_pcd(ca, _demo_points(), np.ones(len(g["point_src"]) + len(g["point_dst"]), dtype=bool))
    nd = len(g["point_dst"])
    want_fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab[nd:], lab[:nd], None, g["gt_flow"])
    want = frame_pairs.register_frame_pair(frame_pairs.default_args(max_points=2048), want_fp, dev)
    assert torch.equal(got["pairs"], want["pairs"]) and torch.equal(got["flow"], want["flow"])
    epe = float(np.linalg.norm(got["flow"].cpu().numpy() - g["gt_flow"], axis=1).mean())
    assert len(got["pairs"]) > 40 and epe < 0.12, (len(got["pairs"]), epe)
    with pytest.raises(ValueError):
        frame_pairs.register_frame_pair(frame_pairs.default_args(max_points=2048), back, dev)


# ============================================================================ HDBSCAN branch (utils_cluster.py:10-29)
from oracle import hdbscan as oh  # noqa: E402


def _partition_mismatch(a, b):
    """points whose cluster differs after matching every cluster of `a` to the cluster of `b` it overlaps most
    (noise is its own class on both sides)"""
    a, b = np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)
    bad = int(((a < 0) != (b < 0)).sum())
    both = (a >= 0) & (b >= 0)
    for c in np.unique(a[both]):
        inside = b[both & (a == c)]
        bad += int(len(inside) - np.bincount(inside).max())
    return bad


def _mismatch_mask(a, b):
    """the points _partition_mismatch counts"""
    a, b = np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)
    bad = (a < 0) != (b < 0)
    both = (a >= 0) & (b >= 0)
    for c in np.unique(a[both]):
        sel = both & (a == c)
        bad |= sel & (b != np.bincount(b[sel]).argmax())
    return bad


def _assert_mismatches_sit_at_tied_merge_heights(ea, eb, w, got, want):
    """Two correct HDBSCAN runs can only differ where the hierarchy is not unique: at TIED merge heights (equal
    mutual-reachability weights, which Prim orders by visiting order, numpy's argsort arbitrarily and this build by
    row numbers).  Checked explicitly: the points labelled differently form connected pieces of the spanning tree,
    and the lightest tree edge leaving each piece -- the height at which it merges with the rest -- has a weight
    that occurs more than once in the tree.  -> number of differing points."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    bad = _mismatch_mask(got, want)
    if not bad.any():
        return 0
    n = len(got)
    vals, counts = np.unique(w, return_counts=True)
    tied = dict(zip(vals.tolist(), (counts > 1).tolist()))
    inside = bad[ea] & bad[eb]
    _, comp = connected_components(coo_matrix((np.ones(int(inside.sum())), (ea[inside], eb[inside])), shape=(n, n)),
                                   directed=False)
    leaving = bad[ea] != bad[eb]                         # edges with exactly one end in the differing set
    piece = np.where(bad[ea], comp[ea], comp[eb])[leaving]
    wl = w[leaving]
    for c in np.unique(comp[bad]):
        lightest = wl[piece == c].min()
        assert tied[float(lightest)], (int((comp == c).sum()), float(lightest))
    return int(bad.sum())


def _lattice(n_side, step=0.25):
    g = np.stack(np.meshgrid(np.arange(n_side), np.arange(n_side), np.arange(3), indexing="ij"), -1).reshape(-1, 3)
    return (g * step).astype(np.float32)


def test_oracle_tree_weights_equal_sklearn_prim():
    """Every minimum spanning tree has the same multiset of weights: the oracle's exact tree against the
    weights of sklearn's Prim tree recorded by the generator (G11), bit for bit."""
    g = load_golden("g11_hdbscan")
    for i in (0, 2):
        p, (k, _) = g[f"crop_{i}_points"], g[f"crop_{i}_params"]
        k = int(k) + 1           # min_cluster_size = min_samples of the hdbscan package: the point itself not counted
        _, _, w2, c2 = oh.mst(p, k)
        assert np.array_equal(np.sort(np.sqrt(w2)), g[f"crop_{i}_tree_weights"])
        from scipy.spatial import cKDTree
        d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
        assert np.array_equal(np.sqrt(c2), d[:, -1])


@pytest.mark.parametrize("case", ["crop_0", "crop_1", "crop_2", "synth"])
def test_host_labels_on_the_oracle_tree_equal_the_reference_run_up_to_tied_heights(case):
    """No GPU: the oracle's exact tree (core distances in the hdbscan package's convention) through the product's
    host routine (icpflow_hdbscan_labels) against the reference's own cluster_pcd run (G11); whatever differs sits
    at a tied merge height."""
    g = load_golden("g11_hdbscan")
    p, (k, ncl), want = g[f"{case}_points"], g[f"{case}_params"], g[f"{case}_labels"]
    mask = g["synth_nonground"] if case == "synth" else np.ones(len(p), dtype=bool)
    ra, rb, rw, _ = oh.mst(p[mask], int(k) + 1)
    lab = _hip().labels_from_mst(ra, rb, np.sqrt(rw), int(mask.sum()), int(k))
    # the fixture went through the reference's keep-the-largest step: compare on the points it kept clustered
    kept = want[mask] >= 0
    n_bad = _assert_mismatches_sit_at_tied_merge_heights(ra, rb, np.sqrt(rw), np.where(kept | (lab < 0), lab, -1), want[mask])
    assert n_bad <= 0.01 * mask.sum()


def _gpu_tree(p, k, mask=None):
    t = _hip().hdbscan_mst(p, k, mask)
    a, b = t["a"].cpu().numpy().astype(np.int64), t["b"].cpu().numpy().astype(np.int64)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    o = np.lexsort((hi, lo))
    return lo[o], hi[o], t["w2"].cpu().numpy()[o], t["core2"].cpu().numpy(), t["n_live"]


@gpu
@pytest.mark.parametrize("case", ["crop_0", "crop_1", "crop_2", "synth", "lattice", "duplicates", "k1", "tiny"])
def test_gpu_spanning_tree_equals_oracle_edge_for_edge(case):
    g = load_golden("g11_hdbscan")
    mask = None
    if case.startswith("crop"):       # k = the package's min_samples + 1 (the kernels count the point itself)
        p, k = g[f"{case}_points"], int(g[f"{case}_params"][0]) + 1
    elif case == "synth":
        p, k, mask = g["synth_points"], int(g["synth_params"][0]) + 1, g["synth_nonground"]
    elif case == "lattice":           # every weight is tied many times over: the tie rule decides everything
        p, k = _lattice(14), 7
    elif case == "duplicates":
        p, k = np.repeat(_cloud(41, 300), 4, axis=0), 6
    elif case == "k1":
        p, k = _cloud(42, 1200), 1
    else:
        p, k = _cloud(43, 9), 3
    lo, hi, w2, c2, n_live = _gpu_tree(p, k, mask)
    sub = p if mask is None else p[mask]
    ra, rb, rw, rc = oh.mst(sub, k)
    rows = np.arange(len(p)) if mask is None else np.flatnonzero(mask)
    assert n_live == len(sub) and len(lo) == len(sub) - 1
    assert np.array_equal(c2[rows], rc) and (mask is None or np.isnan(c2[~mask]).all())
    assert np.array_equal(lo, rows[ra]) and np.array_equal(hi, rows[rb]) and np.array_equal(w2, rw)
    if f"{case}_tree_weights" in g:
        assert np.array_equal(np.sort(np.sqrt(w2)), g[f"{case}_tree_weights"])


@gpu
def test_gpu_spanning_tree_of_the_demo_frame_has_sklearn_prims_weights():
    """126 598 points: the sorted weights of the GPU tree equal those of sklearn's exact Prim tree (G11; three
    minutes on a CPU core), bit for bit; core distances against a KD-tree."""
    g = load_golden("g11_hdbscan")
    pts = _demo_points()
    t = _hip().hdbscan_mst(pts, 21)      # HDBSCAN(min_cluster_size=20, min_samples=None) of the hdbscan package
    assert t["n_live"] == len(pts) and len(t["a"]) == len(pts) - 1
    assert np.array_equal(np.sort(np.sqrt(t["w2"].cpu().numpy())), g["demo_tree_weights"])
    from scipy.spatial import cKDTree
    sample = np.random.default_rng(0).choice(len(pts), 4000, replace=False)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts[sample].astype(np.float64), k=21)
    assert np.array_equal(np.sqrt(t["core2"].cpu().numpy()[sample]), d[:, -1])
    # the edges form one tree
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    a, b = t["a"].cpu().numpy(), t["b"].cpu().numpy()
    nc, _ = connected_components(coo_matrix((np.ones(len(a)), (a, b)), shape=(len(pts), len(pts))), directed=False)
    assert nc == 1


@gpu
@pytest.mark.parametrize("case", ["crop_0", "crop_1", "crop_2", "synth"])
def test_gpu_hdbscan_labels_against_reference_run(case):
    """cluster_pcd with if_hdbscan against the reference's own cluster_pcd (sklearn standing in for the hdbscan
    package): same partition up to points at tied merge heights (Prim's visiting order vs the total order)."""
    g = load_golden("g11_hdbscan")
    p, (k, ncl), want = g[f"{case}_points"], g[f"{case}_params"], g[f"{case}_labels"]
    mask = g["synth_nonground"] if case == "synth" else np.ones(len(p), dtype=bool)
    a = SimpleNamespace(min_cluster_size=int(k), num_clusters=int(ncl), if_hdbscan=True, epsilon=0.25)
    got = _hip().cluster_pcd(a, p, mask)
    assert got.dtype == np.float64 and np.array_equal(got == -1e8, want == -1e8)
    bad = _partition_mismatch(got[mask], want[mask])
    assert bad <= 0.01 * mask.sum(), (bad, int(mask.sum()))
    # the host half on its own: the oracle's tree through the product's host logic gives the product's labels
    ra, rb, rw, _ = oh.mst(p[mask], int(k) + 1)
    lab = _hip().labels_from_mst(ra, rb, np.sqrt(rw), int(mask.sum()), int(k))
    # ... and every point labelled differently from the reference run sits at a tied merge height
    # (the fixture went through the reference's keep-the-largest step: compare on the points it kept clustered)
    kept = want[mask] >= 0
    assert _assert_mismatches_sit_at_tied_merge_heights(ra, rb, np.sqrt(rw), np.where(kept | (lab < 0), lab, -1),
                                                        want[mask]) <= 0.01 * mask.sum()
    full = _hip().hdbscan(p, int(k), None, None if case != "synth" else mask)
    assert np.array_equal(full[mask], lab)


@gpu
def test_gpu_hdbscan_demo_frame_against_reference_run():
    from sklearn.metrics import adjusted_rand_score
    L = load_golden("g8_demo_labels")
    want = np.concatenate([L["label_dst"], L["label_src"]]).astype(np.int64)
    pts = _demo_points()
    a = SimpleNamespace(min_cluster_size=20, num_clusters=200, if_hdbscan=True, epsilon=0.25)
    got = _hip().cluster_pcd(a, pts, np.ones(len(pts), dtype=bool)).astype(np.int64)
    assert adjusted_rand_score(want, got) > 0.999
    assert _partition_mismatch(got, want) < 0.005 * len(pts)
    # the differing points, explicitly: pieces of the spanning tree that merge with the rest at tied heights
    t = _hip().hdbscan_mst(pts, 21)
    raw = _hip().hdbscan(pts, 20)                              # before the keep-largest step (all clusters survive here)
    n_bad = _assert_mismatches_sit_at_tied_merge_heights(t["a"].cpu().numpy().astype(np.int64), t["b"].cpu().numpy().astype(np.int64),
                                                         np.sqrt(t["w2"].cpu().numpy()), raw, want)
    # the gap to the reference run (sklearn's HDBSCAN through the reference's cluster_hdbscan) is 54 points
    # of 126 598 by this count (raw labels, before the keep-largest step; 44 change between noise and cluster in bench.py's
    # count) and one cluster (147 against 148), every one of them at a tied merge height (asserted above).  Bounded HERE at
    # what was measured (+ a handful): a change that widens it fails.
    n_clusters_got, n_clusters_want = int(got.max()) + 1, int(want.max()) + 1
    print(f"HDBSCAN demo frame: {n_bad} points labelled differently, {n_clusters_got} clusters against the reference run's {n_clusters_want}")
    assert n_bad <= 60, n_bad
    assert abs(n_clusters_got - n_clusters_want) <= 1, (n_clusters_got, n_clusters_want)
    with pytest.raises(RuntimeError):
        _hip().hdbscan_mst(pts, 65)                      # min_samples beyond the wave-wide selection
    with pytest.raises(ValueError):
        _hip().hdbscan(pts[:10], 20)


def test_host_tree_labels_equal_sklearns_routines_on_the_same_tree():
    """icpflow_hdbscan_labels (host C++, runs without a GPU) against sklearn's make_single_linkage +
    tree_to_labels fed the same edges in the same order, on the oracle's trees of fixture crops and on random
    trees with heavy ties."""
    import ctypes
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import breadth_first_order, minimum_spanning_tree
    from sklearn.cluster._hdbscan._linkage import MST_edge_dtype, make_single_linkage
    from sklearn.cluster._hdbscan._tree import tree_to_labels
    from icp_flow_amd import _lib

    def ours(a, b, w, n, mcs):
        a, b = np.ascontiguousarray(a, np.int32), np.ascontiguousarray(b, np.int32)
        w = np.ascontiguousarray(w, np.float64)
        out = np.empty(n, np.int32)
        rc = _lib._L.icpflow_hdbscan_labels(a.ctypes.data, b.ctypes.data, w.ctypes.data, n, mcs, out.ctypes.data)
        assert rc == 0
        return out

    def theirs(a, b, w, n, mcs):
        adj = coo_matrix((np.ones(2 * len(a)), (np.r_[a, b], np.r_[b, a])), shape=(n, n)).tocsr()
        _, pred = breadth_first_order(adj, 0, directed=False)
        child_is_a = pred[a] == b
        mst = np.empty(len(a), dtype=MST_edge_dtype)
        mst["current_node"], mst["next_node"], mst["distance"] = np.where(child_is_a, b, a), np.where(child_is_a, a, b), w
        mst = mst[np.lexsort((mst["next_node"], mst["current_node"], mst["distance"]))]
        return np.asarray(tree_to_labels(make_single_linkage(mst), mcs, "eom", False, 0.0, None)[0])

    g = load_golden("g11_hdbscan")
    cases = []
    for i in (0, 1, 2):
        p, k = g[f"crop_{i}_points"][:1200], int(g[f"crop_{i}_params"][0])
        a, b, w2, _ = oh.mst(p, k)
        cases.append((a, b, np.sqrt(w2), len(p), k))
    rng = np.random.default_rng(3)
    for n, mcs, levels in [(2, 2, 3), (5, 2, 2), (60, 3, 4), (400, 5, 6), (1500, 10, 50), (1500, 4, 1000000)]:
        dense = np.triu(rng.integers(1, levels + 1, size=(n, n)).astype(np.float64), 1)   # few distinct weights
        t = minimum_spanning_tree(dense).tocoo()
        perm = rng.permutation(len(t.data))
        cases.append((t.row[perm], t.col[perm], t.data[perm] / levels, n, mcs))
    for a, b, w, n, mcs in cases:
        assert np.array_equal(ours(a, b, w, n, mcs), theirs(a, b, w, n, mcs)), (n, mcs)
    # not a spanning tree / bad arguments
    bad = np.empty(4, np.int32)
    e = np.array([0, 0, 1], np.int32), np.array([1, 1, 0], np.int32), np.ones(3)
    assert _lib._L.icpflow_hdbscan_labels(e[0].ctypes.data, e[1].ctypes.data, e[2].ctypes.data, 4, 2, bad.ctypes.data) != 0


@gpu
def test_gpu_unlabelled_frame_pair_hdbscan_then_registered():
    """The reference's demo pipeline end to end on the GPU: joint HDBSCAN of the stacked frame pair (demo.py:210,
    --if_hdbscan) + track + flow.  Against the reference's own run (G8: labels by sklearn's HDBSCAN, 83 matched
    pairs, EPE 0.0585): the clustering agrees up to tie points, so the matched pairs and the error agree closely."""
    from icp_flow_amd import frame_pairs
    g = load_golden("g8_demo")
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], None, None, None, g["gt_flow"])
    a = frame_pairs.default_args(max_points=2048, cluster="hdbscan", min_cluster_size=20, num_clusters=200)
    got = frame_pairs.register_frame_pair(a, fp, torch.device("cuda", 0))
    flow = got["flow"].cpu().numpy()
    epe = float(np.linalg.norm(flow - g["gt_flow"], axis=1).mean())
    assert abs(len(got["pairs"]) - len(g["pairs"])) <= 3 and abs(epe - float(g["epe"])) < 5e-3, (len(got["pairs"]), epe)
    same = np.abs(flow - g["flow"]).max(axis=1) < 1e-4
    # the reference's stage 1 stopped after 41 iterations (a torch.topk tie, see test_gpu_parity.test_demo_frame_*):
    # the (at most three, large) clusters still moving then are compared in that test, from the reference's own
    # initial poses; every other cluster agrees point by point
    lsrc = load_golden("g8_demo_labels")["label_src"]
    moving = [l for l in np.unique(lsrc[lsrc >= 0]) if (lsrc == l).sum() > 900 and same[lsrc == l].mean() < 0.5]
    rest = ~np.isin(lsrc, moving)
    assert len(moving) <= 5 and rest.mean() > 0.3, moving
    assert same[rest].mean() > 0.97, same[rest].mean()


@gpu
def test_gpu_clustering_input_variants():
    """Rows of four floats (x, y, z, flag) on the GPU, an index list instead of a mask, min_samples different from
    min_cluster_size: same answers as the plain calls / the oracle."""
    hip = _hip()
    p = _cloud(51, 3000)
    wide = torch.from_numpy(np.concatenate([p, np.ones((len(p), 1), np.float32)], 1)).cuda()
    lab3, _ = hip.dbscan(p, 0.3, 6)
    lab4, _ = hip.dbscan(wide, 0.3, 6)
    assert torch.equal(lab3, lab4)
    a = SimpleNamespace(epsilon=0.3, min_cluster_size=6, num_clusters=8, if_hdbscan=False)
    mask = np.random.default_rng(1).random(len(p)) < 0.8
    by_mask = hip.cluster_pcd(a, p, mask)
    by_index = hip.cluster_pcd(a, p, np.flatnonzero(mask))
    assert np.array_equal(by_mask, by_index) and np.array_equal(by_mask, oc.cluster_pcd(a, p, mask))
    # min_samples given explicitly: the tree is the oracle's tree for that k, labels through the host routine
    sub = p[:1500]
    t = hip.hdbscan_mst(torch.from_numpy(sub).cuda(), 7)
    ra, rb, rw, rc = oh.mst(sub, 7)
    lo = np.minimum(t["a"].cpu().numpy(), t["b"].cpu().numpy()).astype(np.int64)
    hi = np.maximum(t["a"].cpu().numpy(), t["b"].cpu().numpy()).astype(np.int64)
    o = np.lexsort((hi, lo))
    assert np.array_equal(lo[o], ra) and np.array_equal(hi[o], rb) and np.array_equal(t["w2"].cpu().numpy()[o], rw)
    got = hip.hdbscan(sub, 25, min_samples=7, counts_self=True)          # scikit-learn's convention
    assert np.array_equal(got, hip.labels_from_mst(ra, rb, np.sqrt(rw), len(sub), 25))
    from sklearn.cluster import HDBSCAN
    ref = HDBSCAN(min_cluster_size=25, min_samples=7, leaf_size=100).fit(sub.astype(np.float64)).labels_
    assert _partition_mismatch(got, ref) <= 0.01 * len(sub)


@gpu
def test_gpu_frame_pairs_cli_clusters_and_registers(tmp_path, capsys):
    """python -m icp_flow_amd.frame_pairs DIR --cluster hdbscan on two unlabelled copies of the demo frame pair."""
    import json
    from icp_flow_amd import frame_pairs
    g = load_golden("g8_demo")
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], None, None, None, g["gt_flow"])
    for k in range(2):
        frame_pairs.save_frame_pair(str(tmp_path / f"f{k}.npz"), fp)
    frame_pairs.main([str(tmp_path), "--max-points", "2048", "--cluster", "hdbscan", "--min-cluster-size", "20",
                      "--num-clusters", "200"])
    s = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert s["frame_pairs"] == 2 and s["evaluated_points"] == 2 * len(g["gt_flow"])
    assert abs(s["epe"] - float(g["epe"])) < 5e-3 and s["ms_per_frame_pair"] > 0
