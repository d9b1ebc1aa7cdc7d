"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Tolerances (BASELINE.json north_star): histogram bins / indices bit-exact; transforms and
per-point motion within 1e-4 m of the reference.
"""
import os

import numpy as np
import pytest
import torch

from conftest import dense_from_sparse, load_golden

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from icp_flow_amd import synthetic  # noqa: E402
from icp_flow_amd import hist as hip_hist  # noqa: E402
from icp_flow_amd import utils_helper, utils_hist, utils_icp, utils_icp_pytorch3d, utils_match  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

from icp_flow_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture(params=["scan", "grid", "sweep"])
def icp_search_mode(request):
    """Every test that runs the ICP loop (marked `all_icp_searches`) runs with all correspondence searches
    of the loop: the all-pairs LDS scan, the exact hashed grid and the sorted sweep must be
    indistinguishable (same tolerances, same iteration counts).  The search is a per-call option
    (icpflow_options_t), in force for the calls of this thread inside the block."""
    with _lib.options(search=request.param):
        yield request.param


all_icp_searches = pytest.mark.usefixtures("icp_search_mode")

TOL_M = 1e-4      # metres, on translations and moved points


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def C(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def moved(T, pts):
    """apply [B,4,4] to [B,N,3] in float64"""
    T = np.asarray(T, np.float64)
    return np.einsum("bij,bnj->bni", T[:, :3, :3], np.asarray(pts, np.float64)) + T[:, None, :3, 3]


def assert_pose_close(got, want, clouds, mask=None, tol=TOL_M):
    """Transforms agree when they move every valid point of the cluster to within tol."""
    got, want = np.asarray(got), np.asarray(want)
    valid = clouds[:, :, 3] > 0
    diff = np.abs(moved(got, clouds[:, :, :3]) - moved(want, clouds[:, :, :3])).max(-1)
    diff = np.where(valid, diff, 0.0).max(1)
    if mask is not None:
        diff = diff[mask]
    assert diff.max() <= tol, f"max point displacement between poses {diff.max():.3e} m (per pair {diff})"


def assert_sqrt_close(got, want):
    """Euclidean distances: the squared distances are bit-identical (checked separately via
    knn_points_lengths); the device sqrt is correctly rounded (as CUDA's sqrtf is), while
    torch-CPU's vectorised sqrt -- used by the oracle and the golden vectors -- is off by
    one ulp on ~1 % of inputs.  Hence: at most 1 ulp."""
    np.testing.assert_array_max_ulp(np.asarray(got, np.float32), np.asarray(want, np.float32), maxulp=1)


# ------------------------------------------------------------------ a-1 vote
def test_hist_known_answer_bit_exact():
    g = load_golden("g1_hist_testpy")
    h = hip_hist.hist(G(g["X"]), G(g["Y"]), *[float(v) for v in g["mins"]], *[float(v) for v in g["maxs"]],
                      *[int(v) for v in g["lens"]])
    want = dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"])
    assert np.array_equal(h.cpu().numpy(), want)
    assert [int(x.argmax()) for x in h] == [111987] * 3          # (50,130,7): X-Y = (-5, 3, 0.2)


@pytest.mark.parametrize("tag", ["tf2p0", "tf3p34"])
def test_hist_reference_style_bit_exact(tag):
    g = load_golden("g1_hist_ref_" + tag)
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    ex, ey, ez = utils_hist.bin_edges(a)
    assert np.array_equal(ex.numpy(), g["edges_x"])
    h = hip_hist.hist(G(g["dst"]), G(g["src"]), ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(),
                      len(ex), len(ey), len(ez))
    want = dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"])
    assert np.array_equal(h.cpu().numpy(), want)


@pytest.mark.parametrize("tf,N", [(2.0, 1024), (6.68, 300), (13.36, 700)])
def test_hist_vs_oracle_random_flags_and_big_histograms(tf, N):
    """LDS-private path (41x41x3) and the global-atomic path (135 / 269 bins per side);
    random (non-prefix) flags like the reference's own test script."""
    S, D, _ = synthetic.make_batch(5, N, seed=101, ragged=True)
    rng = np.random.default_rng(5)
    S[1, :, 3] = rng.integers(0, 2, N)          # interleaved flags
    D[1, :, 3] = rng.integers(0, 2, N)
    a = rp.default_args(translation_frame=tf)
    ex, ey, ez = rp.bin_edges(a)
    args = (ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
    want = rp.hist(C(D), C(S), *args)
    got = hip_hist.hist(G(D), G(S), *args)
    assert np.array_equal(got.cpu().numpy(), want.numpy())
    assert want.sum() > 0


def test_vote_quotient_is_the_ieee_quotient():
    """The vote's (v - min) / (max - min) with the hoisted reciprocal equals the IEEE quotient bit for
    bit: 2^24 numerators per box (dense sweep of [0, r) plus the float neighbours of every bin
    boundary k * r / len), for the boxes of SURVEY A.1 and hist_cuda/test.py."""
    boxes = [(-2.0, 2.0, 41), (-0.1, 0.1, 3), (-3.34, 3.36, 68), (-6.68, 6.72, 135), (-13.36, 13.44, 269),
             (-16.667, 16.733, 335), (-10.0, 10.0, 201), (-0.5, 0.5, 11), (-1.667, 1.733, 35)]
    rng = np.random.default_rng(5)
    for mn, mx, ln in boxes:
        mn32, mx32 = np.float32(mn), np.float32(mx)
        r = np.float32(mx32 - mn32)
        a = (rng.random(1 << 24, dtype=np.float32) * r).astype(np.float32)
        edges = (np.arange(ln + 1, dtype=np.float64) * float(r) / ln).astype(np.float32)
        near = np.concatenate([np.nextafter(edges, np.float32(np.inf)), edges, np.nextafter(edges, np.float32(-np.inf))])
        for _ in range(4):
            near = np.concatenate([near, np.nextafter(near, np.float32(np.inf)), np.nextafter(near, np.float32(-np.inf))])
        a[: len(near)] = np.clip(near, 0, None)
        a[len(near): len(near) + 4] = [0.0, np.nextafter(r, np.float32(0)), 1e-7, 2.4e-7]
        da = G(a)
        fast, ieee = torch.empty_like(da), torch.empty_like(da)
        _lib.call("icpflow_selftest_vote_quotient", _lib.ptr(da), len(a), float(mn32), float(mx32), _lib.ptr(fast),
                  _lib.ptr(ieee), _lib.stream(DEV))
        fast, ieee = fast.cpu().numpy(), ieee.cpu().numpy()
        assert np.array_equal(fast.view(np.uint32), ieee.view(np.uint32)), (mn, mx)
        want = (a.astype(np.float64) / np.float64(r)).astype(np.float32)      # correctly rounded (double quotient
        assert np.array_equal(ieee, want), (mn, mx)                            # of two floats rounds once: 53 >= 2*24+2)


def test_hist_argument_errors_like_the_reference():
    x = torch.zeros(2, 8, 4, device=DEV)
    with pytest.raises(RuntimeError):
        hip_hist.hist(x[:, :, :3], x, -1, -1, -1, 1, 1, 1, 3, 3, 3)          # dim != 4
    with pytest.raises(RuntimeError):
        hip_hist.hist(x, x[:1], -1, -1, -1, 1, 1, 1, 3, 3, 3)                # batch mismatch
    with pytest.raises(RuntimeError):
        hip_hist.hist(x.cpu(), x.cpu(), -1, -1, -1, 1, 1, 1, 3, 3, 3)        # "Not implemented on the CPU"
    with pytest.raises(RuntimeError):
        hip_hist.hist(x.transpose(0, 1), x.transpose(0, 1), -1, -1, -1, 1, 1, 1, 3, 3, 3)   # contiguity


# ------------------------------------------------------------------ a-2 peaks
@pytest.mark.parametrize("tag", ["tf2p0", "tf3p34"])
def test_topk_nms_exact_vs_oracle(tag):
    g = load_golden("g1_hist_ref_" + tag)
    bins = dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"])
    v, i = utils_hist.topk_nms(G(bins))
    wv, wi = rp.topk_nms(C(bins))
    assert np.array_equal(v.cpu().numpy(), wv.numpy())
    assert np.array_equal(i.cpu().numpy(), wi.numpy())
    assert np.array_equal(v.cpu().numpy(), g["peak_votes"])      # reference's vote values


def test_topk_nms_generic_shape_and_kernel():
    rng = np.random.default_rng(3)
    bins = rng.integers(0, 40, size=(3, 23, 17, 11)).astype(np.float32)   # deep z: window smaller than Lz
    for ks in (11, 5, 3):
        v, i = utils_hist.topk_nms(G(bins), k=5, kernel_size=ks)
        wv, wi = rp.topk_nms(C(bins), k=5, kernel_size=ks)
        assert np.array_equal(v.cpu().numpy(), wv.numpy())
        assert np.array_equal(i.cpu().numpy(), wi.numpy())


# ------------------------------------------------------------------ a-4 NN
def test_nearest_neighbor_batch_golden_exact():
    g = load_golden("g3_nn")
    for a, b, tag in ((g["src"], g["dst"], "fwd"), (g["dst"], g["src"], "bwd")):
        idx, dist = utils_helper.nearest_neighbor_batch(G(a), G(b))
        assert np.array_equal(idx.cpu().numpy(), g["idx_" + tag])
        assert_sqrt_close(dist.cpu().numpy(), g["dist_" + tag])


@pytest.mark.parametrize("N,M", [(37, 700), (1500, 900), (2500, 2048)])
def test_nearest_neighbor_batch_vs_oracle_shapes(N, M):
    """ragged sizes, more than one LDS tile, Q=1/2/4 variants, xyz-only (stride 3) input"""
    rng = np.random.default_rng(N + M)
    q = rng.normal(size=(3, N, 4)).astype(np.float32) * 5
    t = rng.normal(size=(3, M, 4)).astype(np.float32) * 5
    t[0, 5] = t[0, 3]                                   # exact duplicate target: first index wins
    q[0, 0, :3] = t[0, 3, :3]
    idx, dist = utils_helper.nearest_neighbor_batch(G(q), G(t))
    widx, wdist = rp.nearest_neighbor_batch(C(q), C(t))
    assert np.array_equal(idx.cpu().numpy(), widx.numpy())
    assert_sqrt_close(dist.cpu().numpy(), wdist.numpy())
    assert int(idx[0, 0]) == 3 and float(dist[0, 0]) == 0.0
    idx3, dist3 = utils_helper.nearest_neighbor_batch(G(q[:, :, :3].copy()), G(t[:, :, :3].copy()))
    assert torch.equal(idx3, idx) and torch.equal(dist3, dist)


def test_knn_with_lengths_matches_pytorch3d_convention():
    S, D, _ = synthetic.make_batch(4, 300, seed=7, ragged=True)
    ls = torch.from_numpy((S[:, :, 3] > 0).sum(1))
    ld = torch.from_numpy((D[:, :, 3] > 0).sum(1))
    d2, idx = utils_helper.knn_points_lengths(G(S), G(D), ls.to(DEV), ld.to(DEV))
    wd2, widx, _ = rp.knn_points(C(S)[:, :, :3], C(D)[:, :, :3], ls, ld)
    assert np.array_equal(idx.cpu().numpy(), widx.numpy())
    assert np.array_equal(d2.cpu().numpy(), wd2.numpy())
    for b in range(4):                                   # rows >= length: idx 0, dist 0
        assert (idx[b, int(ls[b]):] == 0).all() and (d2[b, int(ls[b]):] == 0).all()


def test_transform_points_batch():
    S, _, Tt = synthetic.make_batch(3, 200, seed=9, ragged=True)
    got = utils_helper.transform_points_batch(G(S), G(Tt)).cpu()
    want = rp.transform_points_batch(C(S), C(Tt))
    v = S[:, :, 3] > 0
    np.testing.assert_allclose(got.numpy()[v], want.numpy()[v], atol=2e-5, rtol=0)
    assert np.array_equal(got.numpy()[:, :, 3], S[:, :, 3])


# ------------------------------------------------------------------ a-3 initial pose
def test_estimate_init_pose_vs_oracle_and_golden():
    g = load_golden("g4_init_pose")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    got = utils_hist.estimate_init_pose(a, G(g["src"]), G(g["dst"])).cpu().numpy()
    want = rp.estimate_init_pose(a, C(g["src"]), C(g["dst"])).numpy()
    assert np.array_equal(got, want)                      # same deterministic tie rule: exact
    same_ref = np.abs(got - g["T_init"]).reshape(len(got), -1).max(1) == 0
    assert (same_ref | g["cut_tied"]).all() and same_ref.mean() >= 0.75


# ------------------------------------------------------------------ a-5 ICP
# Tolerances for the ICP family.  The reference (and therefore the oracle and the golden
# vectors) reduces centroids / covariance in fp32 and calls an fp32 SVD; evaluating the SAME
# formulas in fp64 moves its rotation entries by up to ~5e-6 for clusters of a few thousand
# points (measured: test below), i.e. the reference carries that much rounding noise itself.
# The kernels accumulate in fp64, so they are compared (a) tightly against the fp64
# evaluation of the oracle and (b) against the fp32 reference within its own noise; the
# quantity the north star bounds -- where the cluster's points end up -- must agree to 1e-4 m.
TOL_R_REF = 1e-5     # rotation entries vs the fp32 reference / golden
TOL_R_HP = 5e-7      # rotation entries vs the fp64 evaluation of the oracle
TOL_M_HP = 1e-5      # moved points vs the fp64 evaluation (fp32 coordinates at ~50 m: ulp 4e-6)


@all_icp_searches
@pytest.mark.parametrize("case", list("abcde"))
def test_icp_vs_golden(case):
    g = load_golden("g5_icp")
    X, Y = g[case + "_src"], g[case + "_dst"]
    sol = utils_icp_pytorch3d.iterative_closest_point(G(X), G(Y), thres=0.1, max_iterations=100,
                                                      relative_rmse_thr=1e-6)
    R, T = sol.RTs.R.cpu().numpy(), sol.RTs.T.cpu().numpy()
    assert sol.converged.iterations == int(g[case + "_iterations"])
    assert bool(sol.converged) == bool(g[case + "_converged"])
    # < 3 gated correspondences => rank-deficient covariance, rotation backend-dependent
    ok = g[case + "_min_inliers"] >= 3
    if case == "d":      # zero-inlier pair: exact identity, and the batch never "converges"
        assert np.array_equal(R[1], np.eye(3, dtype=np.float32)) and np.array_equal(T[1], np.zeros(3, np.float32))
        assert sol.converged.iterations == 100
        ok[1] = True
    assert ok.sum() >= len(ok) - 1
    np.testing.assert_allclose(R[ok], g[case + "_R"][ok], atol=TOL_R_REF, rtol=0)
    np.testing.assert_allclose(sol.rmse.cpu().numpy()[ok], g[case + "_rmse"][ok], atol=2e-6, rtol=0)
    # y = x R + T on the valid points, against the reference's transformed cloud
    v = (X[:, :, 3] > 0) & ok[:, None]
    np.testing.assert_allclose(sol.Xt.cpu().numpy()[v], g[case + "_Xt"][v], atol=TOL_M, rtol=0)
    assert np.isfinite(R).all() and np.isfinite(T).all()
    np.testing.assert_allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-5)   # proper rotations


@all_icp_searches
@pytest.mark.parametrize("case", list("bce"))
def test_icp_vs_fp64_evaluation_of_the_oracle(case):
    g = load_golden("g5_icp")
    X, Y = g[case + "_src"], g[case + "_dst"]
    want = rp.iterative_closest_point(C(X), C(Y), kabsch_dtype=torch.float64)
    got = utils_icp_pytorch3d.iterative_closest_point(G(X), G(Y))
    ok = g[case + "_min_inliers"] >= 3
    assert got.converged.iterations == want.iterations
    np.testing.assert_allclose(got.RTs.R.cpu().numpy()[ok], want.R.numpy()[ok], atol=TOL_R_HP, rtol=0)
    v = (X[:, :, 3] > 0) & ok[:, None]
    np.testing.assert_allclose(got.Xt.cpu().numpy()[v], want.Xt.numpy()[v], atol=TOL_M_HP, rtol=0)


def test_search_modes_agree():
    """Same gate decisions and neighbours in every mode.  scan and grid visit the source points
    in the same order => bit-identical R, T, rmse; the sweep sums the same fp64 moments in sorted
    order => identical up to the last fp32 bit."""
    S, D, Tt = synthetic.make_batch(12, 700, seed=123, ragged=True)
    src, dst = C(S), C(D)
    for i in range(12):
        v = src[i, :, 3] > 0
        Ti = C(Tt[i])
        src[i, v, 0:3] = src[i, v, 0:3] @ Ti[:3, :3].T + Ti[:3, 3] + torch.tensor([0.04, -0.03, 0.02])
    dst[3, 5] = dst[3, 4]                    # duplicate target points (distance ties)
    dst[3, 40] = dst[3, 4]
    out = {}
    for mode in ("scan", "grid", "sweep"):
        with _lib.options(search=mode):
            sol = utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV))
        out[mode] = (sol.RTs.R.cpu().numpy(), sol.RTs.T.cpu().numpy(), sol.rmse.cpu().numpy(), sol.converged.iterations)
    assert out["scan"][3] == out["grid"][3] == out["sweep"][3]
    for a, b in zip(out["scan"][:3], out["grid"][:3]):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(out["sweep"][0], out["scan"][0], atol=2e-7, rtol=0)
    np.testing.assert_allclose(out["sweep"][1], out["scan"][1], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out["sweep"][2], out["scan"][2], atol=1e-7, rtol=0)


@all_icp_searches
def test_icp_init_transform_and_t_history_vs_oracle():
    """iterative_closest_point(init_transform=...) (utils_icp_pytorch3d.py:118-138) and ICPSolution.t_history (:187):
    the first search runs on X R0 + T0, every iteration's (R, T) comes back.  Against the oracle's trace; a wrong-shaped
    init_transform raises the reference's ValueError, a non-unit scale is refused."""
    S, D, Tt = synthetic.make_batch(10, 400, seed=321, ragged=True, n_min=120)
    src, dst = C(S), C(D)
    # initial transform = the true motion, perturbed by up to 2 degrees about the moved cloud's centre and 6 cm
    M0 = np.empty((10, 4, 4), np.float64)
    for i, deg in enumerate(np.linspace(-2.0, 2.0, 10)):
        c, sn = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        P = np.eye(4)
        P[:3, :3] = [[c, -sn, 0], [sn, c, 0], [0, 0, 1]]
        ctr = (Tt[i].astype(np.float64) @ np.append(S[i, :20, :3].mean(0), 1.0))[:3]
        P[:3, 3] = ctr - P[:3, :3] @ ctr + np.array([0.05, -0.04, 0.01])
        M0[i] = P @ Tt[i].astype(np.float64)
    R0 = C(np.ascontiguousarray(M0[:, :3, :3].transpose(0, 2, 1)).astype(np.float32))   # row convention: y = x R + T
    T0 = C(M0[:, :3, 3].astype(np.float32))
    init = utils_icp_pytorch3d.SimilarityTransform(R0, T0, torch.ones(10))
    want = rp.iterative_closest_point(src, dst, init_transform=(R0, T0), trace=True, kabsch_dtype=torch.float64)
    plain = rp.iterative_closest_point(src, dst, kabsch_dtype=torch.float64)
    got = utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), init_transform=init)
    assert got.converged.iterations == want.iterations
    assert not torch.equal(want.R, plain.R)          # (the initial transform matters on this batch)
    v = S[:, :, 3] > 0
    np.testing.assert_allclose(got.Xt.cpu().numpy()[v], want.Xt.numpy()[v], atol=2e-5, rtol=0)
    hist = got.t_history
    assert len(hist) == want.iterations and len(want.history) == want.iterations
    for k in (0, 1, len(hist) - 1):
        np.testing.assert_allclose(hist[k].R.cpu().numpy(), want.history[k][0].numpy(), atol=2e-6, rtol=0)
        np.testing.assert_allclose(hist[k].T.cpu().numpy(), want.history[k][1].numpy(), atol=2e-4, rtol=0)
    assert torch.equal(hist[-1].R, got.RTs.R) and torch.equal(hist[-1].T, got.RTs.T)
    with pytest.raises(ValueError, match="init_transform"):
        utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), init_transform=(R0[:3], T0, torch.ones(10)))
    # an initial transform with a scale: it shapes the first search only (the alignment returns unit scale unless
    # estimate_scale, utils_icp_pytorch3d.py:376-379)
    s1 = torch.full((10,), 1.002)
    if _lib._current()[-1]["search"] in (0, 3):     # (similarity transforms: the sorted-sweep kernels)
        want_s = rp.iterative_closest_point(src, dst, init_transform=(R0, T0, s1), kabsch_dtype=torch.float64)
        got_s = utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), init_transform=(R0, T0, s1))
        assert got_s.converged.iterations == want_s.iterations
        np.testing.assert_allclose(got_s.Xt.cpu().numpy()[v], want_s.Xt.numpy()[v], atol=2e-5, rtol=0)
        assert bool((got_s.RTs.s == 1).all())
    else:
        with pytest.raises(RuntimeError, match="sorted-sweep"):
            utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), init_transform=(R0, T0, s1))
    # per-pair stop: no per-iteration records -> empty history, like round 1
    assert len(utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), stop_mode="per_pair").t_history) == 0


@pytest.mark.parametrize("B,N", [(8, 400), (6, 1500), (300, 1400), (520, 800), (5, 40)],
                         ids=["one-pass", "teams", "several-passes", "two-workgroups-per-cu", "below-64-points"])
def test_icp_estimate_scale_vs_oracle(B, N):
    """estimate_scale=True (utils_icp_pytorch3d.py:364-374): the transform is a similarity, Xt = s X R + T with
    s = trace(E S) / Xcov.  Targets = the source scaled by 0.9 .. 1.1 about its centre, turned by a few degrees and
    shifted; started from the true scale (init_transform carries s) less a few per cent."""
    rng = np.random.default_rng(21)
    S = np.zeros((B, N, 4), np.float32)
    D = np.zeros((B, N, 4), np.float32)
    s_true = rng.uniform(0.9, 1.1, B)
    for i in range(B):
        p = rng.uniform(-0.5, 0.5, (N, 3)) * np.array([4.0, 1.8, 1.5])
        c = np.array([12.0 + 3 * (i % 16), -7.0 + (i // 16), 0.8])
        yaw = np.deg2rad(rng.uniform(-2, 2))
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        q = s_true[i] * (p @ Rz.T) + c + np.array([0.03, -0.02, 0.01]) + rng.normal(0, 0.003, (N, 3))
        S[i, :, :3], D[i, :, :3] = p + c, q
        S[i, :, 3] = D[i, :, 3] = 1.0
    # initial transform: the true scale within 1 %, about the cluster centre (T0 = c - s0 c)
    s0 = (s_true * rng.uniform(0.99, 1.01, B)).astype(np.float32)
    cen = S[:, :, :3].mean(1)
    R0 = C(np.tile(np.eye(3, dtype=np.float32), (B, 1, 1)))
    T0 = C((cen - s0[:, None] * cen).astype(np.float32))
    want = rp.iterative_closest_point(C(S), C(D), init_transform=(R0, T0, C(s0)), estimate_scale=True,
                                      kabsch_dtype=torch.float64)
    got = utils_icp_pytorch3d.iterative_closest_point(G(S), G(D), init_transform=(R0, T0, C(s0)), estimate_scale=True)
    assert got.converged.iterations == want.iterations
    np.testing.assert_allclose(got.RTs.s.cpu().numpy(), want.s.numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(got.RTs.s.cpu().numpy(), s_true, atol=5e-3)          # and it is the scale that was applied
    np.testing.assert_allclose(got.RTs.R.cpu().numpy(), want.R.numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(got.Xt.cpu().numpy(), want.Xt.numpy(), atol=3e-5, rtol=0)
    np.testing.assert_allclose(got.rmse.cpu().numpy(), want.rmse.numpy(), atol=2e-6)
    assert len(got.t_history) == want.iterations
    np.testing.assert_allclose(got.t_history[-1].s.cpu().numpy(), want.s.numpy(), atol=2e-6)
    # the certificates do not care: identical with the plain scan
    with _lib.options(no_adaptive_windows=True):
        plain = utils_icp_pytorch3d.iterative_closest_point(G(S), G(D), init_transform=(R0, T0, C(s0)), estimate_scale=True)
    assert torch.equal(plain.RTs.R, got.RTs.R) and torch.equal(plain.RTs.s, got.RTs.s) and torch.equal(plain.RTs.T, got.RTs.T)


@all_icp_searches
def test_icp_allow_reflection_vs_oracle():
    """allow_reflection=True (utils_icp_pytorch3d.py:354-362): R = U V^T whatever its determinant.  Targets that are the
    MIRROR image of the source (plus a small rigid motion): the best orthogonal fit is a reflection (det R = -1) with a
    small rmse; without the flag the best rotation is found instead."""
    rng = np.random.default_rng(12)
    B, N = 6, 300
    S = np.zeros((B, N, 4), np.float32)
    D = np.zeros((B, N, 4), np.float32)
    for i in range(B):
        p = rng.normal(size=(N, 3)) * np.array([1.5, 0.8, 0.4]) + np.array([10.0 + i, -5.0, 1.0])
        q = p.copy()
        q[:, 1] = -10.0 - q[:, 1]                                  # mirrored in the plane y = -5
        S[i, :, :3], D[i, :, :3] = p, q + np.array([0.03, -0.02, 0.01]) + rng.normal(0, 0.003, size=(N, 3))
        S[i, :, 3] = D[i, :, 3] = 1.0
    # start from the mirror itself (init_transform): the gate sees neighbours from the first iteration on
    R0 = C(np.tile(np.diag([1.0, -1.0, 1.0]).astype(np.float32), (B, 1, 1)))
    T0 = C(np.tile(np.array([0.0, -10.0, 0.0], np.float32), (B, 1)))
    want = rp.iterative_closest_point(C(S), C(D), init_transform=(R0, T0), allow_reflection=True, kabsch_dtype=torch.float64)
    got = utils_icp_pytorch3d.iterative_closest_point(G(S), G(D), init_transform=(R0, T0, torch.ones(B)), allow_reflection=True)
    R = got.RTs.R.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(np.linalg.det(R), -1.0, atol=1e-5)
    assert got.converged.iterations == want.iterations
    np.testing.assert_allclose(R, want.R.numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(got.Xt.cpu().numpy(), want.Xt.numpy(), atol=2e-5, rtol=0)
    assert float(got.rmse.max()) < 0.01
    # without the flag: a proper rotation, and a much worse fit
    plain = utils_icp_pytorch3d.iterative_closest_point(G(S), G(D), init_transform=(R0, T0, torch.ones(B)))
    np.testing.assert_allclose(np.linalg.det(plain.RTs.R.cpu().numpy().astype(np.float64)), 1.0, atol=1e-5)


@all_icp_searches
def test_icp_per_pair_stop_stays_within_tolerance():
    """Per-pair stopping is NOT the reference's rule: each pair leaves the loop at
    its own convergence instead of iterating until the whole batch satisfies the test.  On
    pairs that really converge the end state must still agree to the 1e-4 m tolerance."""
    g = load_golden("g5_icp")
    X, Y = g["c_src"], g["c_dst"]
    fast = utils_icp_pytorch3d.iterative_closest_point(G(X), G(Y), stop_mode="per_pair")
    v = X[:, :, 3] > 0
    np.testing.assert_allclose(fast.Xt.cpu().numpy()[v], g["c_Xt"][v], atol=TOL_M, rtol=0)
    assert fast.converged.iterations <= int(g["c_iterations"]) and bool(fast.converged)


@all_icp_searches
def test_icp_multi_group_path_vs_oracle():
    """n_src > 2048 exercises the several-query-groups (scratch) path of the ICP kernel."""
    S, D, Tt = synthetic.make_batch(2, 2600, seed=77)
    src, dst = C(S), C(D)
    for i in range(2):                       # pre-align so that ICP has inliers
        Ti = C(Tt[i])
        src[i, :, 0:3] = src[i, :, 0:3] @ Ti[:3, :3].T + Ti[:3, 3] + torch.tensor([0.03, -0.02, 0.01])
    ref = rp.iterative_closest_point(src, dst)
    hp = rp.iterative_closest_point(src, dst, kabsch_dtype=torch.float64)
    got = utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV))
    assert got.converged.iterations == ref.iterations == hp.iterations
    R = got.RTs.R.cpu().numpy()
    np.testing.assert_allclose(R, hp.R.numpy(), atol=TOL_R_HP, rtol=0)
    np.testing.assert_allclose(got.Xt.cpu().numpy(), hp.Xt.numpy(), atol=TOL_M_HP, rtol=0)
    # the fp32 reference sits further from its own fp64 evaluation than the kernel does
    assert np.abs(R - hp.R.numpy()).max() <= np.abs(ref.R.numpy() - hp.R.numpy()).max()
    np.testing.assert_allclose(R, ref.R.numpy(), atol=TOL_R_REF, rtol=0)
    np.testing.assert_allclose(got.Xt.cpu().numpy(), ref.Xt.numpy(), atol=TOL_M, rtol=0)


@all_icp_searches
def test_icp_batch_larger_than_the_gpu_vs_oracle():
    """More pairs than compute units: the single-launch form of the batch-global stop rule has to work
    while late workgroups only start when early pairs have left (periodic pairs publish the rest of their
    history and go; nobody waits).  Same stopping iteration and poses as the oracle, and as the
    one-launch-per-iteration form."""
    B, N = 600, 128
    S, D, Tt = synthetic.make_batch(B, N, seed=31, ragged=True, n_min=40)
    src, dst = C(S), C(D)
    for i in range(B):
        Ti = C(Tt[i])
        src[i, :, 0:3] = torch.where(src[i, :, 3:4] > 0, src[i, :, 0:3] @ Ti[:3, :3].T + Ti[:3, 3] + torch.tensor([0.02, -0.01, 0.01]),
                                     src[i, :, 0:3])
    hp = rp.iterative_closest_point(src, dst, kabsch_dtype=torch.float64, max_iterations=60)
    got = utils_icp_pytorch3d.iterative_closest_point(src.to(DEV), dst.to(DEV), max_iterations=60)
    assert got.converged.iterations == hp.iterations
    inl = (torch.cdist(hp.Xt, dst[:, :, 0:3]).min(dim=2)[0] <= 0.1) & (src[:, :, 3] > 0)
    ok = (inl.sum(dim=1) >= 6).numpy()          # rank-deficient pairs are not pinned
    assert ok.sum() > 0.9 * B
    np.testing.assert_allclose(got.RTs.R.cpu().numpy()[ok], hp.R.numpy()[ok], atol=TOL_R_HP * 4, rtol=0)
    valid = (src[:, :, 3] > 0).numpy()[ok]
    diff = np.abs(got.Xt.cpu().numpy()[ok] - hp.Xt.numpy()[ok]).max(-1)
    assert np.where(valid, diff, 0).max() <= TOL_M


@all_icp_searches
def test_degenerate_pairs_do_not_disturb_their_neighbours():
    """A pair whose src role is empty (all pads) and a pair of two single points share a batch with normal
    pairs: the call returns, the normal pairs come out exactly as without the degenerate ones, the
    degenerate ones are finite-or-NaN but never garbage from a neighbour."""
    S, D, _ = synthetic.make_batch(4, 256, seed=9, ragged=True, n_min=60)
    # per-pair stopping: with the reference's batch-global rule the never-converging degenerate pairs would
    # (correctly) hold the whole batch at the iteration cap and change the neighbours' stopping iteration
    a = rp.default_args(max_points=256, icp_stop_mode="per_pair")
    base = utils_match.hist_icp(a, G(S), G(D)).cpu().numpy()
    S2, D2 = S.copy(), D.copy()
    S2[1, :, 0:3] = 1e8; S2[1, :, 3] = 0.0                       # empty src role
    S2[2, 1:, 0:3] = 1e8; S2[2, 1:, 3] = 0.0                     # one point each
    D2[2, 1:, 0:3] = 1e8; D2[2, 1:, 3] = 0.0
    got = utils_match.hist_icp(a, G(S2), G(D2)).cpu().numpy()
    assert np.array_equal(got[[0, 3]], base[[0, 3]])
    ev = utils_match.match_eval(a, G(S2), G(D2), G(got))
    torch.cuda.synchronize()
    for k in (1, 2):
        T = got[k]
        assert np.all(np.isfinite(T) | np.isnan(T))
    # single point on single point: the histogram proposes the exact translation (on the bin grid) or zero
    assert np.isfinite(got[2]).all() and np.allclose(got[2][:3, :3], np.eye(3), atol=1e-6)


# ------------------------------------------------------------------ a-9 / a-11 / a-12
@all_icp_searches
def test_apply_icp_from_reference_init_poses():
    g = load_golden("g6_hist_icp")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    got, iters = utils_icp.apply_icp(a, G(g["src"]), G(g["dst"]), G(g["T_init_noswap"]), return_iterations=True)
    assert int(iters) == int(g["icp_iterations_noswap"])
    assert_pose_close(got.cpu().numpy(), g["T_apply_icp_noswap"], g["src"])
    rb = g["rolled_back_noswap"]                 # rolled-back pairs return the init pose exactly
    assert np.array_equal(got.cpu().numpy()[rb], g["T_init_noswap"][rb])


@all_icp_searches
def test_rollback_on_identical_clouds():
    g = load_golden("g6_rollback")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    got = utils_match.hist_icp(a, G(g["src"]), G(g["dst"])).cpu().numpy()
    assert np.array_equal(got, g["T_hist_icp"])


@all_icp_searches
def test_hist_icp_ragged_vs_oracle_and_golden():
    g = load_golden("g6_hist_icp")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    got = utils_match.hist_icp(a, G(g["src"]), G(g["dst"])).cpu().numpy()
    want = rp.hist_icp(a, C(g["src"]), C(g["dst"])).numpy()
    assert_pose_close(got, want, g["src"])                            # vs oracle: every pair
    ok = ~g["cut_tied"]
    assert ok.sum() >= 6
    assert_pose_close(got, g["T_hist_icp"], g["src"], mask=ok)        # vs reference: tie-free pairs
    swapped = g["n_src"] > g["n_dst"]
    assert (swapped & ok).any() and (~swapped & ok).any()
    np.testing.assert_array_equal(got[:, 3], np.tile(np.array([0, 0, 0, 1], np.float32), (len(got), 1)))


@all_icp_searches
@pytest.mark.parametrize("tf", [3.34, 6.68, 10.02, 13.36])
def test_hist_icp_larger_translation_frames_vs_oracle(tf):
    """Waymo gap 1 .. gap 4 histogram geometry (main.py:200 at speed 1.67; SURVEY A.1: 68 / 135 / 202 / 269 bins per
    axis, 55 KB .. 868 KB per pair): the vote's bins no longer fit LDS next to the tile (global atomics) and, from
    135 bins on, the peaks kernel works through global scratch volumes.  Initial pose and full registration
    against the oracle."""
    S, D, _ = synthetic.make_batch(6, 256, seed=41, ragged=True, n_min=80)
    a = rp.default_args(max_points=256, translation_frame=tf)
    got0 = utils_hist.estimate_init_pose(a, G(S), G(D)).cpu().numpy()
    want0, aux0 = rp.estimate_init_pose_batch(a, C(S), C(D), return_aux=True)
    tied = (aux0["peak_votes"][:, 4] == 0).numpy()    # fewer than five positive peaks: zero-vote bins complete the
    assert not tied.all()                              # top-5 in an order that is the implementation's, not the data's
    assert np.array_equal(got0[~tied], want0.numpy()[~tied])
    got = utils_match.hist_icp(a, G(S), G(D)).cpu().numpy()
    want = rp.hist_icp(a, C(S), C(D)).numpy()
    assert_pose_close(got, want, S, mask=~tied)


@all_icp_searches
def test_hist_icp_dense_vs_golden_and_match_eval():
    g = load_golden("g6_hist_icp_dense")
    S, D, _ = synthetic.make_batch(int(g["num_pairs"]), int(g["max_points"]), seed=int(g["seed"]))
    a = rp.default_args(max_points=int(g["max_points"]))
    T = utils_match.hist_icp(a, G(S), G(D))
    assert_pose_close(T.cpu().numpy(), g["T_hist_icp"], S)
    ev = utils_match.match_eval(a, G(S), G(D), G(g["T_hist_icp"]))
    for got, key, tol in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations"),
                             (1e-5, 0, 1e-6, 1e-6, 1e-4, 1e-4)):
        np.testing.assert_allclose(got.cpu().numpy(), g["ev_" + key], atol=tol, rtol=1e-5)


def test_match_eval_ragged_vs_golden():
    g = load_golden("g6_hist_icp")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    ev = utils_match.match_eval(a, G(g["src"]), G(g["dst"]), G(g["T_hist_icp"]))
    for got, key, tol in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations"),
                             (1e-5, 0, 1e-6, 1e-6, 1e-4, 1e-4)):
        np.testing.assert_allclose(got.cpu().numpy(), g["ev_" + key], atol=tol, rtol=1e-5)


@all_icp_searches
def test_hist_icp_real_data_shape_large_padding():
    """max_points = 10000 like demo.sh / main.sh: mostly tiny clusters plus a few thousand-point
    ones in the same padded batch (three-level window search, scalar-load sweep, sorts in 128 KiB
    of LDS, several query groups per pair)."""
    N = 10000
    sizes = [(25, 31), (80, 64), (300, 420), (2300, 1900), (6000, 9000), (10000, 10000)]
    S = np.empty((len(sizes), N, 4), np.float32)
    D = np.empty((len(sizes), N, 4), np.float32)
    Tt = np.empty((len(sizes), 4, 4), np.float32)
    for i, (ns, nd) in enumerate(sizes):
        S[i], D[i], Tt[i] = synthetic.make_pair(2 * i, ns, nd, N, seed=900)     # shared-sample pairs
    a = rp.default_args(max_points=N, icp_max_iterations=12)
    got, iters = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    want, aux = rp.hist_icp(a, C(S), C(D), max_iterations=12, return_aux=True)
    assert int(iters) == aux["iterations"]
    assert_pose_close(got.cpu().numpy(), want.numpy(), S, tol=2e-4)   # torch-CPU's long sequential fp32 accumulations
    # the same fp32 oracle with its sums in pairwise order (what a GPU reduction -- the reference's CUDA run -- does):
    # the north-star tolerance holds
    want_tree = rp.hist_icp(a, C(S), C(D), max_iterations=12, sum_order="tree", init=aux["init"])
    assert_pose_close(got.cpu().numpy(), want_tree.numpy(), S, tol=TOL_M)
    ev = utils_match.match_eval(a, G(S), G(D), got)
    wv = rp.match_eval(a, C(S), C(D), got.cpu())
    np.testing.assert_allclose(ev[0].cpu().numpy(), wv[0].numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_array_equal(ev[1].cpu().numpy(), wv[1].numpy())
    np.testing.assert_allclose(ev[2].cpu().numpy(), wv[2].numpy(), atol=1e-6)       # ratios
    np.testing.assert_allclose(ev[3].cpu().numpy(), wv[3].numpy(), atol=1e-6)       # ious
    np.testing.assert_allclose(ev[4].cpu().numpy(), wv[4].numpy(), atol=2e-4)       # translations (fp32 centroids at ~40 m)
    np.testing.assert_allclose(ev[5].cpu().numpy(), wv[5].numpy(), atol=1e-4)       # Euler angles, degrees
    # a non-rigid "transform" (scaled): the backward windows cannot be trusted and must fall back to the
    # whole cloud -- same numbers as the oracle again
    Tn = got.clone()
    Tn[:, 0:3, 0:3] *= 1.05
    ev2 = utils_match.match_eval(a, G(S), G(D), Tn)
    wv2 = rp.match_eval(a, C(S), C(D), Tn.cpu())
    np.testing.assert_allclose(ev2[0].cpu().numpy(), wv2[0].numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_array_equal(ev2[1].cpu().numpy(), wv2[1].numpy())


def test_match_eval_sweeps_on_large_batches_equal_the_scans_and_the_oracle():
    """match_eval runs as sorted sweeps from 1024 points on when the batch is large (api.hip eval_by_sweep; config 2's and
    config 4's shapes), as all-pairs scans otherwise: the nearest-neighbour minima are the same numbers either way, so
    inlier counts are EQUAL and the means differ only in the order of an fp64 sum.  Dense and ragged batches, against the
    scans (`no_eval_sweep`) and, on a sample, against the oracle."""
    for B, N, ragged in ((256, 1024, False), (300, 1500, True), (160, 2048, False)):
        S, D, _ = synthetic.make_batch(B, N, seed=21, ragged=ragged, n_min=20)
        a = rp.default_args(max_points=N, icp_max_iterations=8)
        T = utils_match.hist_icp(a, G(S), G(D))
        ev = [e.cpu().numpy() for e in utils_match.match_eval(a, G(S), G(D), T)]
        with _lib.options(no_eval_sweep=True):
            sv = [e.cpu().numpy() for e in utils_match.match_eval(a, G(S), G(D), T)]
        np.testing.assert_array_equal(ev[1], sv[1])                      # inlier counts
        np.testing.assert_allclose(ev[0], sv[0], atol=1e-7, rtol=1e-6)   # mean errors
        for k in (2, 3, 4, 5):
            np.testing.assert_array_equal(ev[k], sv[k])                  # ratios, ious, translations, rotations
        k = 12
        wv = rp.match_eval(a, C(S[:k]), C(D[:k]), T[:k].cpu())
        np.testing.assert_array_equal(ev[1][:k], wv[1].numpy())
        np.testing.assert_allclose(ev[0][:k], wv[0].numpy(), atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(ev[3][:k], wv[3].numpy(), atol=1e-6)


@all_icp_searches
def test_hist_icp_eval_equals_the_two_calls():
    """icpflow_hist_icp_eval = hist_icp followed by match_eval (utils_match.py:92-93), the metrics taken on the sorted
    clouds the registration left behind (by role: pairs whose dst cloud is the smaller one sit swapped in those arrays).
    Transforms and iteration count bit-identical, inlier counts / ratios / ious / translations / rotations EQUAL, mean
    errors up to the order of an fp64 sum; ragged batches with both kinds of pairs, small and long clouds, the side
    stream switched off (the registration then sorts inside the ICP launch and the metrics count and sort for themselves)."""
    for B, N, kw in ((40, 256, {}), (64, 1024, {}), (24, 3000, {}), (6, 6000, {}), (30, 700, {"no_side_stream": True})):
        S, D, _ = synthetic.make_batch(B, N, seed=31 + N, ragged=True, n_min=20)
        S[::3], D[::3] = D[::3].copy(), S[::3].copy()          # both swap states in one batch
        a = rp.default_args(max_points=N, icp_max_iterations=10)
        with _lib.options(**kw):
            T0, it0 = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
            ev0 = utils_match.match_eval(a, G(S), G(D), T0)
            T1, ev1, it1 = utils_match.hist_icp_eval(a, G(S), G(D), return_iterations=True)
        assert torch.equal(T0, T1) and int(it0) == int(it1)
        ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
        assert (nd < ns).any() and (nd > ns).any()
        np.testing.assert_array_equal(ev1[1].cpu().numpy(), ev0[1].cpu().numpy())
        np.testing.assert_allclose(ev1[0].cpu().numpy(), ev0[0].cpu().numpy(), atol=1e-7, rtol=1e-6)
        for k in (2, 3, 4, 5):
            assert torch.equal(ev1[k], ev0[k])


@all_icp_searches
def test_hist_icp_beyond_the_lds_image_and_beyond_the_sorts():
    """Padded lengths above 12288 (the sorted fixed cloud no longer fits LDS: scalar-load sweep) and above
    16384 (no sorts at all: all-pairs vote, scans and ICP search) must give the registration of the same
    clouds padded to 4096 -- every path is exact, only fp64 summation orders differ."""
    sizes = [(3000, 2600), (700, 900), (4000, 4000)]
    a = rp.default_args(icp_max_iterations=20)
    out = {}
    for N in (4096, 13000, 20000):
        S = np.empty((len(sizes), N, 4), np.float32)
        D = np.empty((len(sizes), N, 4), np.float32)
        for i, (ns, nd) in enumerate(sizes):
            S[i], D[i], _ = synthetic.make_pair(2 * i, ns, nd, N, seed=1234)
        a.max_points = N
        T, it = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
        ev = utils_match.match_eval(a, G(S), G(D), T)
        out[N] = (T.cpu().numpy(), int(it), [e.cpu().numpy() for e in ev], S[:, :4096].copy())
    base = out[4096]
    for N in (13000, 20000):
        T, it, ev, _ = out[N]
        assert it == base[1]
        assert_pose_close(T, base[0], base[3], tol=2e-5)
        np.testing.assert_array_equal(ev[1], base[2][1])                       # inlier counts
        np.testing.assert_allclose(ev[0], base[2][0], atol=1e-6)               # mean errors


@all_icp_searches
def test_hist_icp_under_stream_capture_and_on_two_streams():
    """The fused registration can be captured into a HIP graph (the private side stream of the axis sort
    forks from and joins the capturing stream) and replayed with identical results; two streams running
    registrations concurrently do not share scratch."""
    S, D, _ = synthetic.make_batch(64, 512, seed=3)
    a = rp.default_args(max_points=512, icp_max_iterations=30)
    src, dst = G(S), G(D)
    want = utils_match.hist_icp(a, src, dst).clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            utils_match.hist_icp(a, src, dst)       # warm-up on the capturing stream (allocations, attributes)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = utils_match.hist_icp(a, src, dst)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # the one-call form with the metrics (icpflow_hist_icp_eval), captured and replayed the same way
    want_ev = [e.clone() for e in utils_match.match_eval(a, src, dst, want)]
    with torch.cuda.stream(side):
        utils_match.hist_icp_eval(a, src, dst)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        out2, ev2 = utils_match.hist_icp_eval(a, src, dst)
    out2.zero_()
    for e in ev2:
        e.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2, want)
    assert torch.equal(ev2[1], want_ev[1]) and torch.allclose(ev2[0], want_ev[0], atol=1e-7, rtol=1e-6)
    # two streams, interleaved calls, different inputs
    S2, D2, _ = synthetic.make_batch(64, 512, seed=4)
    src2, dst2 = G(S2), G(D2)
    want2 = utils_match.hist_icp(a, src2, dst2).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            r1 = utils_match.hist_icp(a, src, dst)
        with torch.cuda.stream(s2):
            r2 = utils_match.hist_icp(a, src2, dst2)
        res.append((r1, r2))
    torch.cuda.synchronize()
    for r1, r2 in res:
        assert torch.equal(r1, want) and torch.equal(r2, want2)


# ------------------------------------------------------------------ 8(f): association + flow on the demo frame
def _demo_frame_run(fixture):
    """demo.npz frame pair through the HIP path (match_pcds: both association stages; flow kernel), outputs lined up
    with the pairs of the reference run recorded in `fixture`."""
    from icp_flow_amd import utils_flow, utils_track
    g0 = load_golden("g8_demo")
    g = load_golden(fixture)
    lab = load_golden("g8_demo_labels")
    a = rp.default_args(max_points=int(g["max_points"]), min_cluster_size=20, translation_frame=2.0,
                        thres_box=0.1, thres_rot=0.1, thres_error=0.2, thres_iou=0.2)
    assert len(g["pairs"]) == 83
    torch.manual_seed(0)
    ps, pd = G(g0["point_src"]), G(g0["point_dst"])
    ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
    pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
    flow = utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, torch.eye(4, device=DEV))
    pairs, Tm, flow = pairs.cpu().numpy(), Tm.cpu().numpy(), flow.cpu().numpy()
    ref_pairs = g["pairs"]
    got = {int(p[0]): (int(p[1]), k) for k, p in enumerate(pairs)}
    ref = {int(p[0]): (int(p[1]), k) for k, p in enumerate(ref_pairs)}
    assert set(got) == set(ref), (sorted(set(got) ^ set(ref)))
    assert all(got[s][0] == ref[s][0] for s in ref)
    order = [got[int(p[0])][1] for p in ref_pairs]
    return dict(a=a, g0=g0, g=g, lab=lab, ps=ps, pd=pd, ls=ls, ld=ld, pairs=pairs[order], T=Tm[order], flow=flow)


@all_icp_searches
@pytest.mark.parametrize("fixture", ["g8_demo_cudatopk", "g8_demo_mp10000_cudatopk"])
def test_demo_frame_pair_track_and_flow_vs_reference(fixture):
    """BASELINE config 1 end to end, EVERY point of the frame: demo.npz through the HIP path -- match_pcds (both stages:
    sanity_check, gather / pad, hist_icp, match_eval, reject, arg-min) and the flow kernel -- against the reference's own
    match_pcds + flow_estimation_torch (63 276 points), at max_points 2048 and at the reference's real setting 10000
    (demo.sh:9-13; the 30 000-point wall is subsampled with the reference's own torch.randperm stream, the ICP of the
    large clusters runs as a team of workgroups).

    Tie rule.  `torch.topk` (utils_hist.py:27) cuts the top-5 of the NMS survivors; which of several EQUAL votes make the
    cut is implementation-defined.  The product's rule is (vote descending, flat index ascending), which is what ATen's
    CUDA implementation -- the platform the reference runs on -- yields: its single-block radix select
    (aten/src/ATen/native/cuda/TensorTopK.cu, gatherTopK) finds the k-th value, writes everything strictly greater and
    completes the k outputs with elements equal to it in ascending index order.  These fixtures are the reference's own
    code run with exactly that order in place of torch-CPU's (tools/gen_golden.py `topk_cuda_order`; everything else
    unmodified): stage 1 then runs all 100 iterations (one 21-vs-47-point pair, tied on a vote count of 2 at the cut,
    starts from the zero translation and never has an inlier: rel = NaN, utils_icp_pytorch3d.py:209 cannot fire), exactly
    as here.  No cluster is excused: same 83 matched pairs, per-point flow within 1e-4 m on every point."""
    r = _demo_frame_run(fixture)
    g, flow, pairs = r["g"], r["flow"], r["pairs"]
    assert list(g["stage_iterations"]) == [100, 100]
    err = np.linalg.norm(flow - g["flow"], axis=1)
    lsrc = r["lab"]["label_src"]
    worst = sorted(((float(err[lsrc == p[0]].max()), int(p[0]), int((lsrc == p[0]).sum())) for p in g["pairs"]), reverse=True)[:4]
    print(f"{fixture}: flow vs the reference run (CUDA tie order) max {err.max():.3e} m over {len(err)} points; worst clusters "
          f"(max err, label, points) {worst}")
    assert err.max() < TOL_M, worst
    np.testing.assert_allclose(pairs[:, 2:4], g["pairs"][:, 2:4], atol=2e-4)                 # errors
    np.testing.assert_allclose(pairs[:, 4:6], g["pairs"][:, 4:6], atol=2)                    # inlier counts
    np.testing.assert_allclose(pairs[:, 6:10], g["pairs"][:, 6:10], atol=2e-3)               # ratios, ious
    # the flow kernel alone, fed with the reference's pairs / transforms
    from icp_flow_amd import utils_flow
    flow2 = utils_flow.flow_estimation_torch(r["a"], r["ps"], r["pd"], r["ls"], r["ld"], G(g["pairs"]), G(g["transformations"]),
                                             torch.eye(4, device=DEV))
    np.testing.assert_allclose(flow2.cpu().numpy(), g["flow"], atol=2e-5)
    # EPE against ground truth equals the reference's (utils_eval.py:137-182 epe3d)
    epe = float(np.linalg.norm(flow - r["g0"]["gt_flow"], axis=1).mean())
    assert abs(epe - float(g["epe"])) < 1e-4


@pytest.mark.parametrize("fixture", ["g8_demo", "g8_demo_mp10000"])
def test_demo_frame_pair_vs_the_torch_cpu_tie_order_run(fixture):
    """The same frame against the reference run made with torch-CPU's topk (std::partial_sort: another subset of the
    equal votes): there the tied pair of stage 1 starts from a pose that registers, the batch-global stop fires after 41
    (max_points 2048) / 55 (10000) iterations instead of 100, and the clusters still moving at that iteration end up
    elsewhere -- in the REFERENCE's two runs as much as here: its CUDA-order and CPU-order fixtures differ by up to
    4.5 cm on half of the frame's points.  What can be asserted end to end is therefore exactly that: same matched pairs,
    every cluster on which the reference's two runs agree to 1e-5 m within 1e-4 m here too, and nowhere farther from this
    fixture than the reference's own other run is (plus the tolerance).  Stage by stage, from this run's initial poses,
    every cluster is pinned by test_demo_frame_stages_from_the_reference_initial_poses."""
    r = _demo_frame_run(fixture)
    g, flow = r["g"], r["flow"]
    other = load_golden(fixture + "_cudatopk")["flow"]
    err = np.linalg.norm(flow - g["flow"], axis=1)
    ref_gap = np.linalg.norm(other - g["flow"], axis=1)
    lsrc = r["lab"]["label_src"]
    settled = np.ones(len(err), bool)
    moving = []
    for p in g["pairs"]:
        m = lsrc == p[0]
        if ref_gap[m].max() > 1e-5:
            settled &= ~m
            moving.append((int(p[0]), int(m.sum()), float(ref_gap[m].max()), float(err[m].max())))
    print(f"{fixture}: {len(moving)} clusters on which the reference's own two runs differ (label, points, their gap, ours): {moving}; "
          f"max difference on the other {int(settled.sum())} points {err[settled].max():.3e} m")
    assert err[settled].max() < TOL_M
    assert (err <= ref_gap + TOL_M).all()


@pytest.mark.parametrize("fixture", ["g8_demo", "g8_demo_mp10000"])
def test_demo_frame_stages_from_the_reference_initial_poses(fixture):
    """BASELINE config 1 at max_points 2048 and at the reference's real setting (max_points 10000): both association
    stages of the reference's own run, stage by stage.  The padded batches are rebuilt by the product's gather (same randperm
    stream as the reference: seed 0, source then destination, stage 1 then stage 2), the initial poses of the HIP
    path are compared with the reference's (equal unless the top-5 cut of that pair is tied), and the ICP + roll-back
    (icpflow_apply_icp) runs from the REFERENCE's initial poses: same batch-global iteration count (stage 1: 41 at 2048,
    55 at 10000, converged; stage 2: 100, not converged), and every cluster -- the 10 000-point sample of the 30 000-point wall included -- ends up
    where the reference put it."""
    from icp_flow_amd.utils_check import ClusterTable
    g0, g, lab = load_golden("g8_demo"), load_golden(fixture), load_golden("g8_demo_labels")
    a = rp.default_args(max_points=int(g["max_points"]), min_cluster_size=20, translation_frame=2.0,
                        thres_box=0.1, thres_rot=0.1, thres_error=0.2, thres_iou=0.2)
    ps, pd = G(g0["point_src"]), G(g0["point_dst"])
    ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
    st, dt = ClusterTable.pair(ps, ls, pd, ld)
    torch.manual_seed(0)
    off = 0
    for k, n in enumerate(g["stage_sizes"]):
        pr = g["stage_pairs"][off:off + n]
        ref_init, ref_T, tied = g["stage_init"][off:off + n], g["stage_T"][off:off + n], g["stage_tied"][off:off + n]
        off += n
        si, di = st.find_host(pr[:, 0]), dt.find_host(pr[:, 1])
        S, D = utils_match._gather_pair_batches(a, st, dt, si, di)
        n1, n2 = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
        sw = n1 > n2                                                        # utils_match.py:142
        A = torch.where(sw[:, None, None], D, S).contiguous()
        B = torch.where(sw[:, None, None], S, D).contiguous()
        init = utils_hist.estimate_init_pose(a, A, B).cpu().numpy()
        same = (init == ref_init).all((1, 2))
        assert (~same).sum() <= 1 and not (~same & ~tied).any(), (k, np.nonzero(~same)[0], tied[~same])
        T, iters = utils_icp.apply_icp(a, A, B, G(ref_init), return_iterations=True)
        T = T.cpu().numpy().astype(np.float64)
        swn = sw.cpu().numpy()
        T[swn] = np.linalg.inv(T[swn])                                      # utils_match.py:152-154
        d = np.abs(moved(T, S.cpu().numpy()[:, :, :3]) - moved(ref_T, S.cpu().numpy()[:, :, :3])).max(-1)
        d = np.where((S[:, :, 3] > 0).cpu().numpy(), d, 0.0).max(1)
        finite = np.isfinite(ref_T).all((1, 2))
        print(f"stage {k + 1}: {n} pairs, iterations HIP {int(iters)} reference {int(g['stage_iterations'][k])}, initial poses "
              f"equal on {int(same.sum())}, max displacement {d[finite].max():.3e} m, worst pairs "
              f"{[(int(pr[i, 0]), int(n1[i]), float(d[i])) for i in np.argsort(-d)[:3]]}")
        if int(iters) != int(g["stage_iterations"][k]):
            # The reference's stop at exactly this iteration is decided by fp32 rounding at the 1e-6 threshold (at
            # max_points 2048 the subsampled large clusters never repeat exactly): the exact evaluation of the same
            # formulas (oracle, Kabsch step in fp64, same batch, same initial poses) stops where the kernels do, and
            # every cluster ends up where THAT evaluation puts it.
            M64, aux64 = rp.apply_icp(a, A.cpu(), B.cpu(), C(ref_init), return_aux=True, kabsch_dtype=torch.float64)
            M64 = M64.numpy().astype(np.float64)
            M64[swn] = np.linalg.inv(M64[swn])
            d = np.abs(moved(T, S.cpu().numpy()[:, :, :3]) - moved(M64, S.cpu().numpy()[:, :, :3])).max(-1)
            d = np.where((S[:, :, 3] > 0).cpu().numpy(), d, 0.0).max(1)
            print(f"   exact evaluation of the oracle: {aux64['iterations']} iterations; max displacement HIP vs that {d[finite].max():.3e} m")
            assert int(iters) == aux64["iterations"] and abs(int(iters) - int(g["stage_iterations"][k])) <= 1
        assert d[finite].max() < TOL_M


def test_flow_rigid_with_more_pairs_than_one_label_tile():
    """flow_estimation_torch (utils_flow.py:57-69) with 5000 matched pairs -- the kernel caches the pair labels in
    LDS 2048 at a time; the reference has no such limit -- against the oracle's restatement."""
    from icp_flow_amd import utils_flow
    rng = np.random.default_rng(3)
    P, N = 5000, 60000
    labels = rng.integers(-1, P + 200, N).astype(np.float32)
    labels[rng.random(N) < 0.1] = -1e8
    pts = rng.uniform(-50, 50, (N, 3)).astype(np.float32)
    pair_labels = rng.permutation(P + 200)[:P].astype(np.float32)
    pairs = np.zeros((P, 10), np.float32)
    pairs[:, 0] = pair_labels
    T = np.tile(np.eye(4, dtype=np.float32), (P, 1, 1))
    T[:, :3, 3] = rng.uniform(-1, 1, (P, 3))
    ang = rng.uniform(-0.05, 0.05, P)
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = (0.3, -0.2, 0.01)
    got = utils_flow.flow_estimation_torch(None, G(pts), None, G(labels), None, G(pairs), G(T), G(pose)).cpu().numpy()
    want = rp.flow_estimation_torch(C(pts), C(labels), C(pairs), C(T), C(pose)).numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)
    # the matched labels read in place (column 0 of the [P,10] rows) or from a plain array / another dtype: the same flow
    for other in (G(pairs)[:, 0:1].contiguous(), G(pairs).double(), G(np.concatenate([pairs, pairs], axis=1))[:, :10]):
        again = utils_flow.flow_estimation_torch(None, G(pts), None, G(labels), None, other, G(T), G(pose)).cpu().numpy()
        assert np.array_equal(again, got)
    none = utils_flow.flow_estimation_torch(None, G(pts), None, G(labels), None, G(pairs[:0]), G(T[:0]), G(pose)).cpu().numpy()
    np.testing.assert_allclose(none, pts @ pose[:3, :3].T + pose[:3, 3] - pts, atol=2e-5)


def test_cluster_stats_kernel_vs_torch():
    """icpflow_cluster_stats (centroid, sorted bbox extents per label) against plain torch per cluster."""
    from icp_flow_amd.utils_check import ClusterTable
    rng = np.random.default_rng(12)
    pts = rng.normal(0, 20, size=(20000, 3)).astype(np.float32)
    lab = rng.choice(np.array([-1e8, -1.0, 0, 1, 2, 7, 19, 300], dtype=np.float32), size=len(pts),
                     p=[0.6, 0.1, 0.05, 0.05, 0.1, 0.05, 0.049, 0.001])
    t = ClusterTable(G(pts), G(lab))
    uniq = np.unique(lab)
    assert np.array_equal(t.labels_unq.cpu().numpy(), uniq)
    for k, l in enumerate(uniq):
        sel = pts[lab == l]
        assert int(t.count[k]) == len(sel) == int(t.h_count[k])
        if l < 0:      # ground / noise are never candidates (utils_check.py:32): statistics skipped
            assert not t.mean[k].any() and not t.extent[k].any()
            continue
        np.testing.assert_allclose(t.mean[k].cpu().numpy(), sel.astype(np.float64).mean(0), rtol=0, atol=2e-6)
        want = np.sort(np.abs(sel.max(0) - sel.min(0)))                 # get_bbox_tensor, utils_helper.py:166-170
        assert np.array_equal(t.extent[k].cpu().numpy(), want)


@all_icp_searches
def test_synthetic_frame_pair_vs_oracle():
    """A labelled synthetic frame pair (ragged clusters, relabelled objects -> both association
    stages) through track() + flow kernel against the oracle's match_pcds / flow."""
    from icp_flow_amd import frame_pairs
    d = synthetic.make_frame_pair(seed=2, n_objects=9, n_max=400, n_background=1500)
    fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"])
    a = frame_pairs.default_args(max_points=512)
    out = frame_pairs.register_frame_pair(a, fp, DEV)
    torch.manual_seed(0)
    want_pairs, want_T = rp.match_pcds(a, C(fp.points_src), C(fp.points_dst), C(fp.labels_src), C(fp.labels_dst))
    want_flow = rp.flow_estimation_torch(C(fp.points_src), C(fp.labels_src), want_pairs, want_T, C(fp.pose)).numpy()
    pairs = out["pairs"].cpu().numpy()
    assert np.array_equal(pairs[:, 0:2], want_pairs.numpy()[:, 0:2])
    assert len(pairs) == 9 and (pairs[:, 1] >= 1000).sum() >= 1            # some objects matched by stage 2
    np.testing.assert_allclose(pairs[:, 2:4], want_pairs.numpy()[:, 2:4], atol=2e-4)
    err = np.linalg.norm(out["flow"].cpu().numpy() - want_flow, axis=1)
    assert err.max() < TOL_M, f"per-point flow differs from the oracle by up to {err.max():.3e} m"
    assert np.linalg.norm(out["flow"].cpu().numpy() - fp.gt_flow, axis=1).mean() < 0.02


@pytest.mark.parametrize("mp", [2048, 10000])
def test_batches_padded_to_their_longest_cluster_register_like_max_points_wide_ones(mp):
    """match_pcds pads the candidate batches of a stage to the longest cluster of the stage (utils_match.
    _stage_rows), the reference to max_points (pad_segment, utils_helper.py:185-196).  Padding rows carry flag 0:
    the demo frame pair registered both ways (args.tight_padding False = the reference's width) -- the same matched pairs,
    the same stage-2 draws, transforms and per-point flow within 1e-5 m of each other (the fp64 sums of a pair follow the
    shape of the workgroups that serve it; stage 2 at max_points 10000 is 576 rows wide instead of 10000)."""
    from icp_flow_amd import utils_flow, utils_track, utils_match
    g0, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
    ps, pd = G(g0["point_src"]), G(g0["point_dst"])
    ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
    widths = {}
    orig = utils_match._register_stage

    def spy(args, *rest):
        out = orig(args, *rest)
        widths.setdefault(bool(getattr(args, "tight_padding", True)), []).append(int(out[1].N))
        return out

    runs = {}
    utils_match._register_stage = spy
    try:
        for tight in (True, False):
            a = rp.default_args(max_points=mp, min_cluster_size=20, translation_frame=2.0, thres_box=0.1, thres_rot=0.1,
                                thres_error=0.2, thres_iou=0.2)
            a.tight_padding = tight
            a.device_association = False        # (both stages through _register_stage; the device path has its own test)
            torch.manual_seed(0)
            pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
            flow = utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, torch.eye(4, device=DEV))
            runs[tight] = (pairs.cpu().numpy(), Tm.cpu().numpy(), flow.cpu().numpy())
    finally:
        utils_match._register_stage = orig
    assert widths[False] == [mp, mp] and widths[True][0] == mp and widths[True][1] < mp and widths[True][1] % 64 == 0, widths
    (p1, T1, f1), (p0, T0, f0) = runs[True], runs[False]
    assert np.array_equal(p1[:, 0:2], p0[:, 0:2]) and len(p1) == 83
    np.testing.assert_allclose(T1, T0, atol=1e-5)
    np.testing.assert_allclose(p1[:, 2:4], p0[:, 2:4], atol=1e-5)          # errors
    assert np.array_equal(p1[:, 4:6], p0[:, 4:6])                            # inlier counts
    assert np.linalg.norm(f1 - f0, axis=1).max() < 1e-5


@pytest.mark.parametrize("clusterer", ["dbscan", "hdbscan"])
def test_multi_gap_sequence_vs_oracle(tmp_path, clusterer):
    """BASELINE config 3's shape without the dataset: a 4-frame sample in the reference's Waymo / nuScenes format
    (dataset_pca.py:41-45 keys) through the stream -- per gap j: ego compensation, joint clustering of frame j and
    frame 0 on the GPU, translation_frame = 2 max(speed j, |ego t|) (main.py:200: 68 / 135 / 202 histogram bins per
    axis), both association stages, flow on the raw points with the ego pose composed in (main.py:230-234) --
    against the oracle registering the same clusters with the same per-gap translation frame."""
    from icp_flow_amd import frame_pairs
    d = synthetic.make_sequence(seed=3, num_frames=4, n_objects=9, n_max=500)
    path = os.path.join(tmp_path, "val_seq.npz")
    np.savez(path, **d)
    a = frame_pairs.default_args(max_points=1024, speed=1.67, cluster=clusterer, min_cluster_size=20, range_x=80.0,
                                 range_y=80.0, epsilon=0.8)      # (sparse synthetic shells: ~30 points / m^2)
    fps = frame_pairs.load_any(path, a)
    assert [fp.gap for fp in fps] == [1, 2, 3]
    epes = []
    for fp in fps:
        tf = frame_pairs.frame_translation(a, fp.pose_exact, fp.gap)
        assert tf == max(1.67 * fp.gap, np.linalg.norm(d["ego_motion_gt"][fp.gap][0:3, -1])) * 2            # main.py:200
        out = frame_pairs.register_frame_pair(a, fp, DEV)
        assert out["translation_frame"] == tf
        ps, pd = G(fp.points_src), G(fp.points_dst)
        ls, ld = frame_pairs.cluster_frame_pair(a, ps, pd, fp.nonground_src, fp.nonground_dst)
        oa = rp.default_args(max_points=1024, translation_frame=tf, min_cluster_size=20)
        torch.manual_seed(0)
        want_pairs, want_T = rp.match_pcds(oa, C(fp.points_src), C(fp.points_dst), ls.cpu(), ld.cpu())
        want_flow = rp.flow_estimation_torch(C(fp.points_src_raw), ls.cpu(), want_pairs, want_T, C(fp.pose)).numpy()
        pairs = out["pairs"].cpu().numpy()
        assert np.array_equal(pairs[:, 0:2], want_pairs.numpy()[:, 0:2]) and len(pairs) >= 6
        err = np.linalg.norm(out["flow"].cpu().numpy() - want_flow, axis=1)
        assert err.max() < TOL_M, f"gap {fp.gap}: per-point flow differs from the oracle by up to {err.max():.3e} m"
        # accuracy against the ground truth, on the points of matched clusters and on the static rest (an object the
        # joint clustering split into per-frame fragments stays unmatched: that is the method, not the registration)
        gt_err = np.linalg.norm(out["flow"].cpu().numpy() - fp.gt_flow, axis=1)
        lsn = ls.cpu().numpy()
        settled = np.isin(lsn, pairs[:, 0]) | ~fp.nonground_src
        epes.append(float(gt_err[settled].mean()))
    assert max(epes) < 0.03, epes
    # the same file through the stream harness: three frame pairs, metrics of the reference's evaluation
    s = frame_pairs.run_stream(a, [path], DEV)
    assert s["frame_pairs"] == 3 and s["evaluated_points"] == int((d["time_indice"] > 0).sum()) and s["epe"] < 0.5


def test_frame_pair_stream_on_demo_frame_matches_reference_metrics(tmp_path):
    """The stream harness on BASELINE config 1 (demo frame pair written in the stream format):
    EPE / accuracy metrics equal the reference's compute_epe_test on the reference's own flow (G9)."""
    from icp_flow_amd import frame_pairs
    from icp_flow_amd import utils_eval
    g, lab, g9 = load_golden("g8_demo"), load_golden("g8_demo_labels"), load_golden("g9_epe")
    assert np.allclose(utils_eval.compute_epe_test(g["flow"], g["gt_flow"]), g9["whole"], rtol=0, atol=1e-12)   # the metric code
    # evaluated on the clusters that have settled when the reference's stage 1 stops (the evaluation mask of the
    # stream format): the three large clusters still moving then depend on a torch.topk tie, see
    # test_demo_frame_pair_track_and_flow_vs_reference
    big = [l for l in np.unique(lab["label_src"]) if l >= 0 and (lab["label_src"] == l).sum() > 900]
    mask = (~np.isin(lab["label_src"], big)).astype(np.float32)
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"], mask)
    frame_pairs.save_frame_pair(str(tmp_path / "demo.npz"), fp)
    a = frame_pairs.default_args(max_points=int(g["max_points"]))
    s = frame_pairs.run_stream(a, frame_pairs.list_frame_pairs(str(tmp_path)), DEV)
    assert s["frame_pairs"] == 1 and s["matched_cluster_pairs"] == len(g["pairs"]) and s["evaluated_points"] == int(mask.sum())
    got = np.array([s[m] for m in ("epe", "accs", "accr", "outlier", "Routlier")])
    np.testing.assert_allclose(got, utils_eval.compute_epe_test(g["flow"], g["gt_flow"], mask), rtol=0, atol=5e-4)
    assert s["ms_per_frame_pair"] > 0 and mask.mean() > 0.25


# ------------------------------------------------------------------ full-size properties (BASELINE config 2)
@pytest.fixture(scope="module")
def config2():
    S, D, Tt = synthetic.make_batch(256, 1024, seed=0)
    return S, D, Tt


def test_full_size_hist_vote_count_property(config2):
    """sum(bins[b]) == number of (i,j) whose difference falls inside the box, counted with
    plain torch ops on the device (same fp32 subtract / compare)."""
    S, D, _ = config2
    a = rp.default_args()
    ex, ey, ez = utils_hist.bin_edges(a)
    s, d = G(S[:32]), G(D[:32])
    h = hip_hist.hist(d, s, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
    v = d[:, :, None, 0:3] - s[:, None, :, 0:3]
    lo = torch.tensor([float(ex.min()), float(ey.min()), float(ez.min())], device=DEV)
    hi = torch.tensor([float(ex.max()), float(ey.max()), float(ez.max())], device=DEV)
    inside = ((v >= lo) & (v < hi)).all(-1).sum((1, 2))
    assert torch.equal(h.sum((1, 2, 3)).long(), inside)


def test_full_size_nn_self_match_and_gather_property(config2):
    S, D, _ = config2
    s, d = G(S[:64]), G(D[:64])
    idx, dist = utils_helper.nearest_neighbor_batch(s, s)               # idempotence: own index, 0 m
    assert torch.equal(idx, torch.arange(1024, device=DEV)[None].expand(64, -1)) and float(dist.max()) == 0.0
    idx, dist = utils_helper.nearest_neighbor_batch(s, d)
    nn = torch.gather(d[:, :, 0:3], 1, idx[:, :, None].expand(-1, -1, 3))
    re = (s[:, :, 0:3] - nn).pow(2).sum(-1).sqrt()
    assert torch.allclose(re, dist, atol=1e-6)
    probe = d[:, torch.randint(0, 1024, (64,), device=DEV), 0:3]         # no probe target is closer
    other = (s[:, :, None, 0:3] - probe[:, None]).pow(2).sum(-1).sqrt().min(-1)[0]
    assert bool((dist <= other + 1e-6).all())


@all_icp_searches
def test_full_size_registration_recovers_motion(config2):
    S, D, Tt = config2
    a = rp.default_args(max_points=1024, icp_max_iterations=50)
    T, iters = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    T = T.cpu().numpy()
    assert 1 <= int(iters) <= 50
    err = np.abs(moved(T, S[:, :, :3]) - moved(Tt, S[:, :, :3])).max((1, 2))
    assert np.median(err[0::2]) < 0.005 and err[0::2].max() < 0.02      # shared-sample pairs
    assert np.isfinite(T).all()
    # oracle on a bounded sample of the same batch (first 4 pairs), same iteration cap
    want = rp.hist_icp(a, C(S[:4]), C(D[:4]), max_iterations=50).numpy()
    got = utils_match.hist_icp(a, G(S[:4]), G(D[:4])).cpu().numpy()
    assert_pose_close(got, want, S[:4])


def test_full_size_init_pose_pick_is_the_argmin_of_all_six_scores(config2):
    """BASELINE config 2 size: the branch-and-bound scoring picks the candidate a plain evaluation of ALL twelve
    means picks (torch ops on the device: exact fp32 differences, every scan to the end), for every pair whose
    two best scores are not within rounding of each other."""
    S, D, _ = config2
    a = rp.default_args()
    ex, ey, ez = utils_hist.bin_edges(a)
    nb = 64
    s, d = G(S[:nb]), G(D[:nb])
    T = utils_hist.estimate_init_pose(a, s, d)
    h = hip_hist.hist(d, s, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
    _, idx = utils_hist.topk_nms(h)
    H, W, Dz = len(ex), len(ey), len(ez)
    exd, eyd, ezd = ex.to(DEV), ey.to(DEV), ez.to(DEV)
    t = torch.stack([exd[idx // Dz // W % H], eyd[idx // Dz % W], ezd[idx % Dz]], dim=-1)      # utils_hist.py:78
    t = torch.cat([t, t.new_zeros(nb, 1, 3)], dim=1)                                            # + zero translation
    scores = torch.empty(nb, 6, device=DEV)
    for k in range(6):
        moved = s[:, :, 0:3] + t[:, k, None, :]
        diff = moved[:, :, None, :] - d[:, None, :, 0:3]
        d2 = diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2]
        fwd = d2.min(dim=2).values.sqrt().mean(dim=1)
        bwd = d2.min(dim=1).values.sqrt().mean(dim=1)
        scores[:, k] = torch.minimum(fwd, bwd)
    best2 = scores.topk(2, dim=1, largest=False).values
    clear = (best2[:, 1] - best2[:, 0]) > 1e-5 * best2[:, 1]
    pick = scores.argmin(dim=1)
    want = t[torch.arange(nb, device=DEV), pick]
    assert int(clear.sum()) >= nb * 3 // 4
    assert torch.equal(T[clear][:, 0:3, 3], want[clear])
    eye = torch.eye(3, device=DEV).expand(nb, 3, 3)
    assert torch.equal(T[:, 0:3, 0:3], eye)


def test_frame_pairs_in_flight_equal_one_after_the_other(tmp_path):
    """frame_pairs.register_in_flight: several frame pairs at once (one HIP stream each, asynchronous device -> host
    hand-overs into pinned memory, one host thread resuming whichever has landed; team launches chained by an event).
    Frame pairs are independent (main.py:184-215): every one comes out exactly as from register_frame_pair with the same
    association path -- pairs, transforms and per-point flow bit for bit -- whatever runs next to it; labelled synthetic pairs of different sizes,
    the demo frame pair (teams of workgroups on its large clusters, twice, so that two team launches are in flight), and a
    multi-gap sequence whose pairs are clustered on the GPU.  Then the stream harness with in_flight = 3."""
    from icp_flow_amd import frame_pairs
    fps = []
    for k, (nobj, nmax) in enumerate(((9, 400), (14, 900), (5, 2500), (11, 300), (7, 1500))):
        d = synthetic.make_frame_pair(seed=20 + k, n_objects=nobj, n_max=nmax, n_background=1000)
        fps.append(frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"]))
    g0, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
    demo = frame_pairs.FramePair(g0["point_src"], g0["point_dst"], lab["label_src"], lab["label_dst"], None, g0["gt_flow"])
    fps = fps[:2] + [demo] + fps[2:] + [demo]
    a = frame_pairs.default_args(max_points=4096)
    # Three hosts: the native one (icpflow_track_frame, a host thread per frame pair in flight: the default), the Python
    # scheduler with the device-side association, the Python scheduler with the host-side association.  With any of them a
    # frame pair comes out bit for bit the same in flight and on its own; the first two agree bit for bit with each other,
    # the host-side association with both to rounding (same pairs).
    wants = {}
    for name, native, assoc, flights in (("native", True, None, (2, 4)), ("device", False, True, (3,)), ("host", False, False, (2, 4))):
        a.native_host, a.device_association = native, assoc
        wants[name] = want = [frame_pairs.register_frame_pair(a, fp, DEV) for fp in fps]
        for in_flight in flights:
            got = {}
            for idx, fp, out in frame_pairs.register_in_flight(a, fps, DEV, in_flight=in_flight):
                assert fp is fps[idx]
                got[idx] = out
            torch.cuda.synchronize()
            assert sorted(got) == list(range(len(fps)))
            for k, w in enumerate(want):
                for key in ("pairs", "transformations", "flow"):
                    assert torch.equal(got[k][key], w[key]), (name, in_flight, k, key)
    for k in range(len(fps)):
        for key in ("pairs", "transformations", "flow"):
            assert torch.equal(wants["native"][k][key], wants["device"][k][key]), (k, key)
        assert torch.equal(wants["device"][k]["pairs"][:, :2], wants["host"][k]["pairs"][:, :2])
        assert (wants["device"][k]["flow"] - wants["host"][k]["flow"]).abs().max() < 1e-5
    a.native_host, a.device_association = True, None
    # the harness: same accuracy summary as one at a time, on files (a sequence file among them: GPU clustering per gap)
    paths = []
    for k, fp in enumerate(fps[:4]):
        paths.append(os.path.join(tmp_path, f"pair{k}.npz"))
        frame_pairs.save_frame_pair(paths[-1], fp)
    seq = synthetic.make_sequence(seed=3, num_frames=3, n_objects=6, n_max=400)
    paths.append(os.path.join(tmp_path, "val_seq.npz"))
    np.savez(paths[-1], **seq)
    a2 = frame_pairs.default_args(max_points=2048, speed=1.67, cluster="dbscan", min_cluster_size=20, epsilon=0.8)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")           # (the synthetic sequence only carries ground-truth ego poses)
        one = frame_pairs.run_stream(a2, paths, DEV)
        many = frame_pairs.run_stream(a2, paths, DEV, in_flight=3)
    assert many["frame_pairs"] == one["frame_pairs"] == 6 and many["matched_cluster_pairs"] == one["matched_cluster_pairs"]
    assert many["evaluated_points"] == one["evaluated_points"] and abs(many["epe"] - one["epe"]) < 1e-6   # (device / host association)


def test_cluster_table_chain_equals_the_torch_ops():
    """icpflow_cluster_table (key kernel, radix sort, boundaries, rows, statistics: one chain of launches) against the
    chain of torch ops it replaces (stable argsort, unique_consecutive, cumsum + icpflow_cluster_stats): the same row
    order, labels, counts, starts, centroids and extents, bit for bit; float labels of either sign, a cloud with more
    distinct labels than the device table holds (falls back to the torch ops), a single-cluster cloud."""
    from icp_flow_amd.utils_check import ClusterTable, TABLE_ROWS
    rng = np.random.default_rng(5)
    cases = {"frame": rng.choice(np.array([-1e8, -1.0, 0, 1, 2, 7, 19, 300, 2.5, -3.25], dtype=np.float32), size=70000,
                                 p=[0.5, 0.1, 0.05, 0.05, 0.1, 0.05, 0.05, 0.04, 0.03, 0.03]),
             "many": rng.integers(0, TABLE_ROWS - 3, 50000).astype(np.float32),
             "overflow": rng.integers(0, TABLE_ROWS + 500, 60000).astype(np.float32),
             "one": np.full(777, 4.0, np.float32)}
    for name, lab in cases.items():
        pts = rng.normal(0, 20, size=(len(lab), 3)).astype(np.float32)
        t = ClusterTable(G(pts), G(lab))
        ref = ClusterTable.__new__(ClusterTable)
        ref.points, ref._lab = t.points, G(lab)
        ref._from_torch()
        assert torch.equal(t.order, ref.order), name
        for attr in ("h_labels", "h_count", "h_start", "h_mean", "h_extent"):
            assert np.array_equal(getattr(t, attr), getattr(ref, attr)), (name, attr)
        assert torch.equal(t.labels_unq, ref.labels_unq) and torch.equal(t.mean, ref.mean) and torch.equal(t.extent, ref.extent)
        assert np.array_equal(t.h_labels, np.unique(lab))
    st, dt = ClusterTable.pair(G(pts), G(cases["one"]), G(pts[:500]), G(cases["one"][:500]))
    assert list(st.h_count) == [777] and list(dt.h_count) == [500]


def test_cluster_table_pair_chain_equals_two_single_chains():
    """icpflow_cluster_table_pair (both clouds of a frame pair through ONE sort with the cloud's number above the label's
    bits) against two icpflow_cluster_table calls: row order, labels, counts, starts, centroids, extents and the number of
    clusters bit for bit -- clouds of different sizes, labels of either sign shared and not shared between the two, one
    cloud overflowing the device table (its table falls back to the torch ops, the other one stays), single-row clouds."""
    from icp_flow_amd.utils_check import ClusterTable, TABLE_ROWS
    rng = np.random.default_rng(6)
    vals = np.array([-1e8, -1.0, 0, 1, 2, 7, 19, 300, 2.5, -3.25], dtype=np.float32)
    prob = [0.5, 0.1, 0.05, 0.05, 0.1, 0.05, 0.05, 0.04, 0.03, 0.03]
    cases = {"frames": (rng.choice(vals, size=70000, p=prob), rng.choice(vals[1:], size=41234)),
             "many": (rng.integers(0, TABLE_ROWS - 3, 50000).astype(np.float32), rng.integers(-5, 40, 3000).astype(np.float32)),
             "overflow in one": (rng.integers(0, 50, 20000).astype(np.float32), rng.integers(0, TABLE_ROWS + 500, 60000).astype(np.float32)),
             "single rows": (np.array([3.0], np.float32), np.array([-2.0], np.float32)),
             "same label everywhere": (np.full(5000, 9.0, np.float32), np.full(64, 9.0, np.float32))}
    for name, (la, lb) in cases.items():
        pa = rng.normal(0, 20, size=(len(la), 3)).astype(np.float32)
        pb = rng.normal(0, 20, size=(len(lb), 3)).astype(np.float32)
        st, dt = ClusterTable.pair(G(pa), G(la), G(pb), G(lb))
        for got, pts, lab in ((st, pa, la), (dt, pb, lb)):
            want = ClusterTable(G(pts), G(lab))
            assert torch.equal(got.order, want.order), name
            for attr in ("h_labels", "h_count", "h_start", "h_mean", "h_extent"):
                assert np.array_equal(getattr(got, attr), getattr(want, attr)), (name, attr)
            assert torch.equal(got.labels_unq, want.labels_unq) and torch.equal(got.mean, want.mean) and torch.equal(got.extent, want.extent)
            assert np.array_equal(got.h_labels, np.unique(lab)), name


def test_match_pcds_with_an_empty_cloud_returns_no_pairs():
    """a frame whose points were all filtered out (an empty cloud on either side) is "no pairs", like the
    reference's match_pcds on empty label sets -- not an error from the cluster-table kernels; the flow of such a frame pair
    is the ego flow alone."""
    from icp_flow_amd import frame_pairs
    a = frame_pairs.default_args(max_points=512)
    pts = torch.randn(300, 3, device=DEV)
    lab = torch.zeros(300, device=DEV)
    empty_p, empty_l = torch.zeros((0, 3), device=DEV), torch.zeros((0,), device=DEV)
    for sp, sl, dp, dl in ((empty_p, empty_l, pts, lab), (pts, lab, empty_p, empty_l), (empty_p, empty_l, empty_p, empty_l)):
        pairs, T = utils_match.match_pcds(a, sp, dp, sl, dl)
        assert tuple(pairs.shape) == (0, 10) and tuple(T.shape) == (0, 4, 4)


@pytest.mark.parametrize("seed,n_objects,n_max,max_points", [(2, 9, 400, 512), (5, 14, 300, 512), (11, 6, 700, 1024), (3, 10, 400, 128)],
                         ids=["two-stages", "more-objects", "larger", "over-long-clusters-fall-back"])
def test_device_association_equals_the_host_path(seed, n_objects, n_max, max_points):
    """match_pcds with both stages enqueued from the cluster tables on (icpflow_assoc_assign / _collect, stage 2 as a superset
    with options.d_pair_active: utils_match._match_pcds_device, the default) against the host path (args.device_association =
    False: every stage's results read back, numpy in between).  Relabelled objects, so that stage 2 has work: the same matched
    pairs in the same order, errors / inliers / transforms to rounding (stage 2's batch is wider and larger on the device path:
    other workgroup shapes, same registrations), per-point flow within 1e-5 m.  With max_points below the cluster sizes the
    relabelled clusters are over-long AND unmatched after stage 1: their random subsamples must be drawn in the reference's
    order, the device path gives up and the host path answers -- bit for bit."""
    from icp_flow_amd import frame_pairs, utils_flow
    d = synthetic.make_frame_pair(seed=seed, n_objects=n_objects, n_max=n_max, n_background=1200)
    ps, pd, ls, ld = G(d["points_src"]), G(d["points_dst"]), G(d["labels_src"]).float(), G(d["labels_dst"]).float()
    pose = G(d["pose"])
    out = {}
    tried = []
    orig = utils_match._match_pcds_device

    def spy(*a, **k):
        r = yield from orig(*a, **k)
        tried.append(r is not None)
        return r

    utils_match._match_pcds_device = spy
    try:
        for device_path in (True, False):
            a = frame_pairs.default_args(max_points=max_points)
            a.device_association = device_path
            a.native_host = False                     # (the Python host's two paths; the native call has its own test)
            a.generator = torch.Generator()
            a.generator.manual_seed(0)
            pairs, T = utils_match.match_pcds(a, ps, pd, ls, ld)
            flow = utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, T, pose)
            out[device_path] = (pairs.cpu().numpy(), T.cpu().numpy(), flow.cpu().numpy())
    finally:
        utils_match._match_pcds_device = orig
    (pd_, Td, fd), (ph, Th, fh) = out[True], out[False]
    assert len(tried) == 1                                   # the device path was entered once (and not at all when switched off)
    assert pd_.shape == ph.shape and np.array_equal(pd_[:, :2], ph[:, :2])
    assert len(ph) >= 3
    if max_points == 128:
        assert tried == [False]                              # gave up: an over-long cluster needed its second try
        assert np.array_equal(pd_, ph) and np.array_equal(Td, Th) and np.array_equal(fd, fh)
        return
    assert tried == [True]
    assert (ph[:, 1] >= 1000).sum() >= 1                     # some objects matched by stage 2
    np.testing.assert_allclose(pd_[:, 2:4], ph[:, 2:4], atol=2e-5)          # errors
    assert np.array_equal(pd_[:, 4:6], ph[:, 4:6])                           # inlier counts
    np.testing.assert_allclose(Td, Th, atol=2e-5)
    assert np.abs(fd - fh).max() < 1e-5


def test_pairs_outside_the_batch_do_not_touch_its_stop_rule():
    """options.d_pair_active: pairs flagged 0 (handed over as empty clouds) are not in the batch -- the registrations and the
    iteration count of the flagged pairs are those of the batch made of the flagged pairs alone (to rounding: the launch shape
    differs), whatever sits in the other rows; refused outside the single speculative launch of the reference stop."""
    S, D, _ = synthetic.make_batch(40, 512, seed=77, ragged=True, n_min=40)
    keep = np.zeros(40, dtype=bool)
    keep[[1, 4, 5, 9, 17, 18, 30, 39]] = True
    S2, D2 = S.copy(), D.copy()
    S2[~keep] = 0; S2[~keep, :, :3] = 1e8
    D2[~keep] = 0; D2[~keep, :, :3] = 1e8
    a = rp.default_args(max_points=512, icp_max_iterations=100)
    T_sub, it_sub = utils_match.hist_icp(a, G(S[keep]), G(D[keep]), return_iterations=True)
    with _lib.options(pair_active=G(keep.astype(np.uint8))):
        T_all, it_all = utils_match.hist_icp(a, G(S2), G(D2), return_iterations=True)
    assert int(it_all) == int(it_sub)
    np.testing.assert_allclose(T_all.cpu().numpy()[keep], T_sub.cpu().numpy(), atol=2e-6)
    # without the mask the empty pairs never satisfy the stop test: the batch runs to the cap
    _, it_nomask = utils_match.hist_icp(a, G(S2), G(D2), return_iterations=True)
    assert int(it_nomask) == 100 > int(it_sub)
    with _lib.options(pair_active=G(keep.astype(np.uint8))):
        with pytest.raises(RuntimeError, match="d_pair_active"):
            utils_match.hist_icp(rp.default_args(max_points=512, icp_stop_mode="per_pair"), G(S2), G(D2))


@pytest.mark.parametrize("case", ["synthetic", "draws", "demo-2048", "demo-10000", "fallback", "stage2-teams"])
def test_native_frame_pair_equals_the_python_host(case):
    """icpflow_track_frame (frame_pairs.register_frame_pair_native: the host half of match_pcds in C++ -- cluster tables,
    candidate lists, sanity_check, padded batches with the stream of random subsamples restated on MT19937, stage 2's superset,
    the checks at the end) against the Python host with the device-side association: pairs, transforms and per-point flow bit
    for bit.  "draws": clusters longer than max_points in stage 1 (torch.randperm's draws).  "fallback": an over-long cluster
    needs its second try -- the superset falls short, the call registers the reference's exact stage 2 on top of stage 1 (the
    Python host goes through its host-side association: the same bits)."""
    from icp_flow_amd import frame_pairs
    if case.startswith("demo"):
        g0, lab = load_golden("g8_demo"), load_golden("g8_demo_labels")
        fps = [frame_pairs.FramePair(g0["point_src"], g0["point_dst"], lab["label_src"], lab["label_dst"], None, g0["gt_flow"])]
        a = frame_pairs.default_args(max_points=int(case.split("-")[1]))
    elif case == "stage2-teams":
        # a few LARGE clusters most of which change their label -- stage 2 is a batch of a handful of pairs wider than
        # 1024 points, i.e. its ICP takes teams, whose plan (the order of a team's sums) follows the pairs that are in the batch:
        # the overlapped mode (the whole superset, the mask known only afterwards) must plan as the serial one does
        fps = []
        for k, (nobj, nmin, nmax) in enumerate(((6, 1400, 3200), (8, 1100, 2600), (5, 2000, 3200))):
            d = synthetic.make_frame_pair(seed=90 + k, n_objects=nobj, n_min=nmin, n_max=nmax, relabel=0.7, n_background=800)
            fps.append(frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"]))
        a = frame_pairs.default_args(max_points=4096)
    else:
        fps = []
        for k, (nobj, nmax) in enumerate(((9, 400), (14, 900), (5, 2500), (11, 300))):
            d = synthetic.make_frame_pair(seed=40 + k, n_objects=nobj, n_max=nmax, n_background=1000)
            fps.append(frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"]))
        if case == "synthetic":
            a = frame_pairs.default_args(max_points=4096)
        else:
            # the background cluster (1000 points) and the larger objects are over-long: subsampled in stage 1
            a = frame_pairs.default_args(max_points=512 if case == "draws" else 128)
    a.device_association = True
    served = 0
    for fp in fps:
        a.native_host = False               # (the Python host: generators, device-side association, its own fall-back)
        want = frame_pairs.register_frame_pair(a, fp, DEV)
        a.native_host = True
        got = frame_pairs.register_frame_pair_native(a, fp, DEV)
        torch.cuda.synchronize()
        # where the Python host's device path gives up (an over-long cluster needs its second try) the call registers the exact
        # stage 2 itself: either way the same bits as the Python host
        assert frame_pairs._served(got), case
        served += want["association"] == "device"
        assert len(want["pairs"]) >= 3
        for key in ("pairs", "transformations", "flow"):
            assert torch.equal(got[key], want[key]), (case, key, want["association"])
        # stage 2's initial poses behind stage 1 on the caller's stream (ICPFLOW_OPT_NO_STAGE_OVERLAP) instead of beside stage 1's ICP
        # on a second one (the default above): the same bits
        a.stage_overlap = False
        serial = frame_pairs.register_frame_pair_native(a, fp, DEV)
        a.stage_overlap = None
        torch.cuda.synchronize()
        for key in ("pairs", "transformations", "flow"):
            assert torch.equal(serial[key], want[key]), (case, key, "stage_overlap=False")
    if case == "fallback":
        assert served < len(fps)            # at least one frame pair needed the exact stage 2
    elif not case.startswith("demo") and case != "stage2-teams":   # (stage2-teams: whichever association the Python host took, the three runs agree)
        assert served >= 1


def test_match_pcds_native_call_continues_torchs_generators():
    """utils_match.match_pcds goes through icpflow_track_frame with the state of the generator the Python host would draw from
    -- `args.generator`, else torch's global one -- and hands the advanced state back: same pairs and transforms as the Python
    host bit for bit, and the generator ends in the same state (over-long clusters, so that there are draws; two calls in a
    row, so that the second starts mid-stream)."""
    from icp_flow_amd import frame_pairs
    fps = []
    for k, (nobj, nmax) in enumerate(((9, 400), (14, 900))):
        d = synthetic.make_frame_pair(seed=60 + k, n_objects=nobj, n_max=nmax, n_background=1000)
        fps.append([G(d["points_src"]), G(d["points_dst"]), G(d["labels_src"]).float(), G(d["labels_dst"]).float()])
    a = frame_pairs.default_args(max_points=384)
    a.device_association = True

    def both_calls(native, generator):
        a.native_host = native
        a.generator = generator
        out = [utils_match.match_pcds(a, *fp) for fp in fps]
        torch.cuda.synchronize()
        return out

    saved = torch.get_rng_state()
    try:
        results = {}
        for native in (True, False):
            torch.manual_seed(5)
            results[native, "global"] = (both_calls(native, None), torch.get_rng_state().clone())
            g = torch.Generator()
            g.manual_seed(9)
            results[native, "own"] = (both_calls(native, g), g.get_state().clone())
    finally:
        torch.set_rng_state(saved)
    for which in ("global", "own"):
        (out_n, state_n), (out_p, state_p) = results[True, which], results[False, which]
        assert torch.equal(state_n, state_p), which
        for (pn, Tn), (pp, Tp) in zip(out_n, out_p):
            assert len(pp) >= 3 and torch.equal(pn, pp) and torch.equal(Tn, Tp), which
    # (and the draws mattered: the two generators give different subsamples, hence different numbers somewhere)
    assert not all(torch.equal(x[1], y[1]) for x, y in zip(results[True, "global"][0], results[True, "own"][0]))
