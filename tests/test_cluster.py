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
    with pytest.raises(NotImplementedError):
        hip.cluster_pcd(SimpleNamespace(if_hdbscan=True), p, np.ones(len(p), bool))
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
