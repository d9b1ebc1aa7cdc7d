"""The oracle (oracle/reference_path.py + oracle_core.c) against the golden vectors
produced by the reference's own Python (tools/gen_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import dense_from_sparse, load_golden
from oracle import reference_path as rp
from icp_flow_amd import synthetic

T = torch.from_numpy


def test_hist_known_answer_of_reference_test_script():
    """hist_cuda/test.py: X-Y = (-5, 3, 0.2) must peak at bin (50,130,7) = flat 111987."""
    g = load_golden("g1_hist_testpy")
    h = rp.hist(T(g["X"]), T(g["Y"]), *g["mins"], *g["maxs"], *[int(v) for v in g["lens"]])
    want = dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"])
    assert np.array_equal(h.numpy(), want)
    assert [int(x.argmax()) for x in h] == [111987] * 3
    assert [float(x.max()) for x in h] == [535.0, 503.0, 508.0]
    lx, ly, lz = (int(v) for v in g["lens"])
    a = 111987
    assert (a // lz // ly % lx, a // lz % ly, a % lz) == (50, 130, 7)


def test_hist_reference_style_calls_bit_exact():
    for tag in ("tf2p0", "tf3p34"):
        g = load_golden("g1_hist_ref_" + tag)
        a = rp.default_args(translation_frame=float(g["translation_frame"]))
        ex, ey, ez = rp.bin_edges(a)
        assert np.array_equal(ex.numpy(), g["edges_x"]) and np.array_equal(ez.numpy(), g["edges_z"])
        assert [len(ex), len(ey), len(ez)] == [int(v) for v in g["lens"]]
        h = rp.hist(T(g["dst"]), T(g["src"]), ex.min(), ey.min(), ez.min(), ex.max(), ey.max(),
                    ez.max(), len(ex), len(ey), len(ez))
        want = dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"])
        assert np.array_equal(h.numpy(), want)
        # hand-made border pair: on-min votes, on-max does not, duplicates add up
        assert h[5].sum() == want[5].sum() and want[5].max() == 2.0


def test_topk_nms_matches_reference_on_tie_free_entries():
    """torch.topk orders equal votes arbitrarily (see the golden idx rows); the oracle's
    rule is (vote desc, flat index asc).  Values must always agree; indices must agree
    wherever the vote is unique among all surviving peaks of that pair."""
    checked = 0
    for tag in ("tf2p0", "tf3p34"):
        g = load_golden("g1_hist_ref_" + tag)
        bins = T(dense_from_sparse(g["bins_shape"], g["bins_nz"], g["bins_val"]))
        votes, idx = rp.topk_nms(bins)
        rv, ri = g["peak_votes"], g["peak_idx"]
        assert np.array_equal(votes.numpy(), rv)
        survivors = rp.nms_mask(bins).reshape(len(rv), -1)
        for b in range(len(rv)):
            for k in range(rv.shape[1]):
                if rv[b, k] > 0 and int((survivors[b] == rv[b, k]).sum()) == 1:
                    assert int(idx[b, k]) == int(ri[b, k])
                    checked += 1
            # the oracle's own order is deterministic: ties by ascending flat index
            for k in range(rv.shape[1] - 1):
                if votes[b, k] == votes[b, k + 1]:
                    assert int(idx[b, k]) < int(idx[b, k + 1])
    assert checked >= 10


def test_nearest_neighbor_batch():
    g = load_golden("g3_nn")
    for a, b, tag in ((g["src"], g["dst"], "fwd"), (g["dst"], g["src"], "bwd")):
        idx, dist = rp.nearest_neighbor_batch(T(a), T(b))
        assert np.array_equal(idx.numpy(), g["idx_" + tag])
        assert np.array_equal(dist.numpy(), g["dist_" + tag])


def _pairs_equal(got, want, atol):
    return np.abs(np.asarray(got) - np.asarray(want)).reshape(len(got), -1).max(1) <= atol


def test_estimate_init_pose():
    """Exact where the top-5 cut is unambiguous; where the 5th/6th peak tie on a positive
    vote count (flag computed by the generator) torch.topk's arbitrary tie order may pick
    other candidates, so only those pairs may differ."""
    g = load_golden("g4_init_pose")
    a = rp.default_args(translation_frame=float(g["translation_frame"]), chunk_size=5)
    Tm = rp.estimate_init_pose(a, T(g["src"]), T(g["dst"]))
    same = _pairs_equal(Tm.numpy(), g["T_init"], 0.0)
    assert (same | g["cut_tied"]).all()
    assert same.mean() >= 0.75


def test_iterative_closest_point_all_cases():
    g = load_golden("g5_icp")
    for k in "abcde":
        sol = rp.iterative_closest_point(T(g[k + "_src"]), T(g[k + "_dst"]), trace=True)
        assert sol.iterations == int(g[k + "_iterations"]), k
        assert bool(sol.converged) == bool(g[k + "_converged"])
        np.testing.assert_allclose(sol.R.numpy(), g[k + "_R"], atol=2e-6)
        np.testing.assert_allclose(sol.T.numpy(), g[k + "_T"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(sol.rmse.numpy(), g[k + "_rmse"], atol=1e-6)
        # whole trajectory, not only the end point
        hR = np.stack([h[0].numpy() for h in sol.history])
        np.testing.assert_allclose(hR, g[k + "_hist_R"], atol=2e-6)
    # degenerate batch: zero-inlier pair returns identity and blocks the global stop
    assert int(g["d_iterations"]) == 100 and not bool(g["d_converged"])
    np.testing.assert_array_equal(g["d_R"][1], np.eye(3, dtype=np.float32))
    np.testing.assert_array_equal(g["d_T"][1], np.zeros(3, np.float32))


def test_tree_order_and_fp64_evaluations_are_the_pinned_algorithm():
    """The two auxiliary evaluations of the oracle the full-size GPU tests use to find out which pairs the
    reference's arithmetic pins -- fp32 sums in pairwise order (`sum_order="tree"`), Kabsch step in fp64
    (`kabsch_dtype`) -- are the same algorithm: on the small golden cases (where rounding cannot decide anything)
    they reproduce the reference's own run like the plain restatement does."""
    g = load_golden("g5_icp")
    x = torch.arange(1, 2001, dtype=torch.float32).reshape(1, 1000, 2) * 0.37
    np.testing.assert_allclose(rp.tree_sum(x, 1).numpy(), x.double().sum(1).numpy(), rtol=1e-6)
    assert rp.tree_sum(torch.ones(3, 5, 4), 1).tolist() == [[5.0] * 4] * 3           # zero padding to 8 rows
    for k in "abce":
        for kw in (dict(sum_order="tree"), dict(kabsch_dtype=torch.float64)):
            sol = rp.iterative_closest_point(T(g[k + "_src"]), T(g[k + "_dst"]), **kw)
            assert sol.iterations == int(g[k + "_iterations"]), (k, kw)
            # pairs with a rank-deficient covariance (< 3 gated correspondences) have no unique rotation: the
            # reference's answer there is an accident of its SVD backend -- excluded, like in the GPU tests
            ok = g[k + "_min_inliers"] >= 3
            v = (g[k + "_src"][:, :, 3] > 0) & ok[:, None]
            np.testing.assert_allclose(sol.Xt.numpy()[v], g[k + "_Xt"][v], atol=2e-5, rtol=0)


def test_hist_icp_and_match_eval_ragged():
    g = load_golden("g6_hist_icp")
    a = rp.default_args(translation_frame=float(g["translation_frame"]), chunk_size=4)
    src, dst = T(g["src"]), T(g["dst"])
    assert (g["n_src"] > g["n_dst"]).any(), "fixture must contain swapped pairs"
    Tm, aux = rp.hist_icp(a, src, dst, return_aux=True)
    same = _pairs_equal(Tm.numpy(), g["T_hist_icp"], 1e-5)
    assert (same | g["cut_tied"]).all() and same.mean() >= 0.75
    assert same[g["n_src"] > g["n_dst"]].any(), "a swapped (inverted) pair must be pinned"
    assert aux["iterations"] >= 2
    init = rp.estimate_init_pose(a, src, dst)
    same_i = _pairs_equal(init.numpy(), g["T_init_noswap"], 0.0)
    assert (same_i | g["cut_tied_noswap"]).all() and same_i.mean() >= 0.75
    # apply_icp from the REFERENCE's init poses: no tie dependence left
    Ti, aux2 = rp.apply_icp(a, src, dst, T(g["T_init_noswap"]).clone(), return_aux=True)
    np.testing.assert_allclose(Ti.numpy(), g["T_apply_icp_noswap"], atol=1e-5)
    assert aux2["iterations"] == int(g["icp_iterations_noswap"])
    assert np.array_equal(aux2["rolled_back"].numpy(), g["rolled_back_noswap"])
    ev = rp.match_eval(a, src, dst, T(g["T_hist_icp"]))
    for got, key in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations")):
        np.testing.assert_allclose(got.numpy(), g["ev_" + key], atol=1e-5, rtol=1e-5)


def test_rollback_on_identical_clouds():
    """e_icp >= e_init (utils_icp.py:34): identical clouds keep the histogram's pose."""
    g = load_golden("g6_rollback")
    a = rp.default_args(translation_frame=float(g["translation_frame"]))
    Tm, aux = rp.hist_icp(a, T(g["src"]), T(g["dst"]), return_aux=True)
    assert bool(aux["rolled_back"].all())
    assert np.array_equal(Tm.numpy(), g["T_hist_icp"]) and np.array_equal(g["T_hist_icp"], g["T_init"])


def test_hist_icp_dense_config2_shape():
    g = load_golden("g6_hist_icp_dense")
    S, D, Tt = synthetic.make_batch(int(g["num_pairs"]), int(g["max_points"]), seed=int(g["seed"]))
    assert np.array_equal(Tt, g["T_true"]) and not g["cut_tied"].any()
    a = rp.default_args(max_points=int(g["max_points"]))
    Tm = rp.hist_icp(a, T(S), T(D))
    np.testing.assert_allclose(Tm.numpy(), g["T_hist_icp"], atol=1e-5)
    ev = rp.match_eval(a, T(S), T(D), T(g["T_hist_icp"]))
    for got, key in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations")):
        np.testing.assert_allclose(got.numpy(), g["ev_" + key], atol=1e-5, rtol=1e-5)
    # sanity of the workload: on the shared-sample (even) pairs the reference recovers the
    # synthetic motion to < 1 cm everywhere on the cluster
    p = S[:, :, 0:3]
    ref = np.einsum("bij,bnj->bni", g["T_hist_icp"][:, :3, :3], p) + g["T_hist_icp"][:, None, :3, 3]
    tru = np.einsum("bij,bnj->bni", Tt[:, :3, :3], p) + Tt[:, None, :3, 3]
    assert np.abs(ref - tru).max(axis=(1, 2))[0::2].max() < 0.01


import pytest


@pytest.mark.parametrize("name", ["g12_config2", "g12_config4_sample"])
def test_headline_sizes_against_the_reference_import_run(name):
    """G12: the reference's own utils_match.hist_icp + match_eval (ICP capped at 50 iterations, the cap
    of BASELINE configs 2 and 4) on the WHOLE config-2 batch (256 x 1024) and on config 4's 64-pair sample (64 x 2048) --
    the sizes the bench is quoted on.  Initial poses bit for bit, the batch-global iteration count equal, transforms and
    metrics within 1e-5 (in the generating container the restatement's transforms equal the run's bit for bit; the tolerance
    leaves room for another host's BLAS rounding the [n,3]x[3,3] products differently, reference_path.point_mm).  The
    fixture also records that the reference's result did not depend on the torch thread count (1 / 8 / 32) there."""
    g = load_golden(name)
    B, N = int(g["num_pairs"]), int(g["max_points"])
    S, D, _ = synthetic.make_batch(int(g["make_batch_pairs"]), N, seed=int(g["seed"]))
    S, D = S[:B], D[:B]
    assert not g["cut_tied"].any()
    torch.set_num_threads(int(g["torch_threads"]))
    a = rp.default_args(max_points=N, icp_max_iterations=int(g["icp_max_iterations"]))
    Tm, aux = rp.hist_icp(a, T(S), T(D), max_iterations=int(g["icp_max_iterations"]), return_aux=True)
    assert np.array_equal(aux["init"].numpy(), g["T_init"])
    assert aux["iterations"] == int(g["icp_iterations"])
    print(f"{name}: restatement == reference-import run bit for bit: {np.array_equal(Tm.numpy(), g['T_hist_icp'])}, "
          f"max |dT| {np.abs(Tm.numpy() - g['T_hist_icp']).max():.2e}, {aux['iterations']} iterations")
    np.testing.assert_allclose(Tm.numpy(), g["T_hist_icp"], atol=1e-5)
    ev = rp.match_eval(a, T(S), T(D), T(g["T_hist_icp"]))
    for got, key in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations")):
        np.testing.assert_allclose(got.numpy(), g["ev_" + key], atol=1e-5, rtol=1e-5)
    for n in (1, 32):
        assert int(g[f"alt{n}_iterations"]) == int(g["icp_iterations"]) and len(g[f"alt{n}_idx"]) == 0


def _stage_batch(a, ps, pd, ls, ld, pairs):
    """The padded batch the reference's match_pairs registers (utils_match.py:81-91) and its smaller-cloud-first
    arrangement (utils_match.py:139-146)."""
    S, D = [], []
    for p in pairs:          # source then destination, pair by pair: the reference's stream of randperm draws
        S.append(rp.pad_segment(ps[ls == p[0], 0:3], a.max_points))
        D.append(rp.pad_segment(pd[ld == p[1], 0:3], a.max_points))
    S, D = torch.stack(S), torch.stack(D)
    sw = (S[:, :, -1] > 0).sum(1) > (D[:, :, -1] > 0).sum(1)
    A, B = S.clone(), D.clone()
    A[sw], B[sw] = D[sw], S[sw]
    return S, A, B, sw


def test_demo_frame_pair_match_pcds_and_flow():
    """G8 (BASELINE config 1): the oracle's restatement of match_pcds (both association stages, sanity_check,
    reject + row arg-min) and of flow_estimation_torch on the reference's demo frame, against the reference's own
    run (83 matched clusters, per-point flow of 63 276 points).

    One thing in that run is not a portable expectation: a candidate pair with fewer than five positive
    vote peaks gets its top-5 completed with zero-vote bins in torch.topk's implementation-defined order, and the
    batch-global ICP stop (utils_icp_pytorch3d.py:209) couples every pair of the batch to that pick: in the
    reference's run stage 1 stops after 41 iterations, with the deterministic tie rule of the restatement (vote
    desc, index asc) the picked pose has no inlier, rel = NaN, and the batch runs all 100 -- which moves the (few,
    large) clusters that are still moving at iteration 41.  So: (1) per stage, from the reference's OWN initial
    poses the restatement reproduces the reference bit for bit -- iteration count and every transform; (2) end to
    end, the same pairs are matched and every cluster that has settled by then agrees exactly."""
    g = load_golden("g8_demo")
    lab = load_golden("g8_demo_labels")
    a = rp.default_args(max_points=int(g["max_points"]), min_cluster_size=20, translation_frame=2.0,
                        thres_box=0.1, thres_rot=0.1, thres_error=0.2, thres_iou=0.2)
    ps, pd = T(g["point_src"]), T(g["point_dst"])
    ls, ld = T(lab["label_src"]).float(), T(lab["label_dst"]).float()
    # (1) stage by stage from the reference's initial poses
    torch.manual_seed(0)
    off = 0
    for k, n in enumerate(g["stage_sizes"]):
        pr, ref_init, ref_T = g["stage_pairs"][off:off + n], T(g["stage_init"][off:off + n]), g["stage_T"][off:off + n]
        tied = g["stage_tied"][off:off + n]
        off += n
        S, A, B, sw = _stage_batch(a, ps, pd, ls, ld, pr)
        init = rp.estimate_init_pose(a, A, B)
        same = (init == ref_init).all(-1).all(-1).numpy()
        assert (~same).sum() <= 1 and not (~same & ~tied).any(), (k, np.nonzero(~same)[0])
        M, aux = rp.apply_icp(a, A, B, ref_init, return_aux=True)
        assert aux["iterations"] == int(g["stage_iterations"][k]) and aux["converged"] == bool(g["stage_converged"][k])
        M[sw] = torch.linalg.inv(M[sw])                                       # utils_match.py:152-154
        np.testing.assert_array_equal(M.numpy(), ref_T)
    # (2) end to end
    torch.manual_seed(0)
    pairs, Tm = rp.match_pcds(a, ps, pd, ls, ld)
    ref_pairs, ref_T = g["pairs"], g["transformations"]
    assert np.array_equal(pairs[:, 0:2].numpy(), ref_pairs[:, 0:2])
    settled = np.abs(Tm.numpy() - ref_T).max((1, 2)) <= 1e-6
    assert (~settled).sum() <= 3, np.nonzero(~settled)[0]
    np.testing.assert_allclose(pairs.numpy()[settled], ref_pairs[settled], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(pairs.numpy()[:, 2:4], ref_pairs[:, 2:4], atol=5e-3)        # errors of the moving ones
    flow = rp.flow_estimation_torch(ps, ls, pairs, Tm, torch.eye(4)).numpy()
    on_settled = np.isin(lab["label_src"], ref_pairs[settled, 0]) | ~np.isin(lab["label_src"], ref_pairs[:, 0])
    np.testing.assert_allclose(flow[on_settled], g["flow"][on_settled], atol=1e-6)
    assert np.abs(flow - g["flow"]).max() < 0.05
    # flow kernel restatement alone, from the reference's pairs / transforms: exact
    flow2 = rp.flow_estimation_torch(ps, ls, T(ref_pairs), T(ref_T), torch.eye(4))
    np.testing.assert_allclose(flow2.numpy(), g["flow"], atol=1e-6)
