"""BASELINE's headline shapes through the C ABI against the oracle on the WHOLE batch.

  * config 2 (256 cluster pairs x 1024 points, <= 50 ICP iterations, the reference's batch-global stop):
    every pair of the batch against `rp.hist_icp` on the same batch -- initial poses, iteration count,
    where the points end up (north_star: < 1e-4 m).
  * the bins of the FUSED vote (the z-sorted kernel the bench times), bit for bit against the oracle.
  * config 4's per-GPU shape (1024 pairs x 2048 points): size-independent properties on the whole shard,
    the oracle on a 64-pair batch of it.

Which pairs can the oracle pin?  The reference's tensors are fp32 and its answer for a pair that is still
moving when the batch rule stops -- and the iteration at which the batch rule stops -- depend on the rounding of
its own reductions, i.e. on the SUMMATION ORDER of the backend it runs on.  Measured: the fp32
oracle on this very batch stops after 47 iterations with 8 torch threads and runs into the cap of 50 with 32
(torch-CPU splits its sums by thread count); with its sums in pairwise order (`sum_order="tree"`, what a GPU
reduction does -- and the reference runs on CUDA) it stops after 47, like its exact evaluation
(`kabsch_dtype=torch.float64`: same formulas, same fp32 inputs) and like the HIP path.  So the test takes THREE
evaluations of the oracle -- fp32 in torch's order, fp32 in pairwise order, Kabsch step in fp64 -- and calls a pair
DETERMINED when all three agree to DETERMINED_TOL.  On every determined pair the HIP path must meet the north-star
tolerance against the fp32 oracle in torch's order -- the reference's arithmetic as restated.  The mask is computed
here, from the oracle alone, and its size is bounded, so a regression cannot hide in it.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from icp_flow_amd import _lib, synthetic, utils_hist, utils_match  # noqa: E402
from oracle import core as ocore  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

DEV = torch.device("cuda:0")
TOL_M = 1e-4             # north_star: flow error vs reference < 1e-4 m
DETERMINED_TOL = 2e-5    # the oracle's fp32 and fp64-Kabsch evaluations agree to this => the oracle pins the pair


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def C(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def displacement(Ta, Tb, clouds):
    """Per pair: the largest distance (max over valid points and axes) between where Ta and Tb put a point."""
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
    p = clouds[:, :, :3].astype(np.float64)
    ma = np.einsum("bij,bnj->bni", Ta[:, :3, :3], p) + Ta[:, None, :3, 3]
    mb = np.einsum("bij,bnj->bni", Tb[:, :3, :3], p) + Tb[:, None, :3, 3]
    d = np.abs(ma - mb).max(-1)
    return np.where(clouds[:, :, 3] > 0, d, 0.0).max(1)


def _all_host_threads():
    # 32 threads: the many small torch ops of the oracle stop scaling there (256 threads: 80x SLOWER, measured)
    n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    ocore.set_num_threads(n)


def _oracle_three(args, S, D, cap):
    """The oracle in the reference's fp32 (torch's summation order), in fp32 with pairwise sums, and with its
    Kabsch step in fp64 -- all from the same initial poses.  -> dict of (T [B,4,4] numpy, aux) + the mask of
    pairs on which the three agree to DETERMINED_TOL."""
    _all_host_threads()
    T32, aux32 = rp.hist_icp(args, C(S), C(D), max_iterations=cap, return_aux=True)
    Ttr, auxtr = rp.hist_icp(args, C(S), C(D), max_iterations=cap, return_aux=True, sum_order="tree", init=aux32["init"])
    T64, aux64 = rp.hist_icp(args, C(S), C(D), max_iterations=cap, return_aux=True, kabsch_dtype=torch.float64,
                             init=aux32["init"])
    T32, Ttr, T64 = T32.numpy(), Ttr.numpy(), T64.numpy()
    self_dis = np.maximum.reduce([displacement(T32, T64, S), displacement(Ttr, T64, S), displacement(T32, Ttr, S)])
    return dict(T32=T32, aux32=aux32, Ttr=Ttr, auxtr=auxtr, T64=T64, aux64=aux64, determined=self_dis < DETERMINED_TOL,
                iters=f"fp32 oracle (torch order, {torch.get_num_threads()} threads) {aux32['iterations']}, fp32 oracle "
                      f"(pairwise order) {auxtr['iterations']}, fp64-Kabsch oracle {aux64['iterations']}")


# ------------------------------------------------------------------------------------------ config 2
@pytest.fixture(scope="module")
def config2():
    S, D, _ = synthetic.make_batch(256, 1024, seed=0)
    args = rp.default_args(max_points=1024, icp_max_iterations=50)
    return dict(S=S, D=D, args=args, **_oracle_three(args, S, D, 50))


def _report(name, err, ok):
    return (f"{name}: {int((err < TOL_M).sum())}/{len(err)} pairs within {TOL_M:g} m, determined {int(ok.sum())}, "
            f"max on determined {err[ok].max():.2e}, max overall {err.max():.2e}, worst {np.argsort(-err)[:6].tolist()}")


def test_config2_full_batch_initial_poses_equal_the_oracle(config2):
    c = config2
    init = utils_hist.estimate_init_pose(c["args"], G(c["S"]), G(c["D"])).cpu().numpy()
    assert np.array_equal(init, c["aux32"]["init"].numpy())


def test_config2_full_batch_vs_oracle(config2):
    """Default arithmetic (fp64 moments).  Iteration count = the oracle's when its sums are exact or pairwise;
    every determined pair within 1e-4 m of the fp32 oracle; the undetermined pairs are few and are exactly the
    ones on which the oracle disagrees with itself."""
    c = config2
    T, iters = utils_match.hist_icp(c["args"], G(c["S"]), G(c["D"]), return_iterations=True)
    T = T.cpu().numpy()
    assert np.isfinite(T).all()
    determined = c["determined"]
    err32 = displacement(T, c["T32"], c["S"])
    errtr = displacement(T, c["Ttr"], c["S"])
    err64 = displacement(T, c["T64"], c["S"])
    msg = "\n".join([_report("HIP vs fp32 oracle (torch order)", err32, determined),
                     _report("HIP vs fp32 oracle (pairwise order)", errtr, determined),
                     _report("HIP vs fp64-Kabsch oracle", err64, determined),
                     f"iterations HIP {int(iters)}, " + c["iters"],
                     f"oracle self-disagreement > {DETERMINED_TOL:g} m on pairs {np.nonzero(~determined)[0].tolist()}"])
    print(msg)
    assert int(iters) == c["aux64"]["iterations"] == c["auxtr"]["iterations"], msg
    # EVERY pair, no mask: the whole 47-iteration trajectory against the exact (fp64 Kabsch) evaluation of the oracle
    # (measured 2.4e-7 m; with the oracle's small matrix products pinned -- reference_path.point_mm -- its moved points
    # are the kernels' bit for bit, so nothing but the Kabsch arithmetic separates the two)
    assert err64.max() < 1e-5, msg
    # the reference's own fp32 arithmetic (torch's and pairwise summation order): the north-star bound on every pair it
    # pins; a pair outside the bound must be one of the oracle's own undetermined pairs (a report on ITS rounding:
    # every pair is pinned by the line above and, step by step against the fp32 oracle, by tests/test_gpu_onestep.py)
    # Measured on the MI355X box's host (32 threads): 247 determined pairs, the other nine are
    # [47, 69, 88, 115, 127, 129, 169, 182, 215]; four of them (215, 47, 69, 129) end up more than 1e-4 m from the
    # torch-order fp32 evaluation (5.4 mm at most), three from the pairwise-order one (0.52 mm at most) -- and the two
    # fp32 evaluations of the oracle are as far from each other.  The bounds below are those counts + 2 and hard ceilings
    # on what the excused pairs may hide (COVERAGE.md "named deviations"; the reference's CUDA tree reductions are a third
    # fp32 evaluation of the same trajectory).
    undetermined = np.nonzero(~determined)[0]
    print("excused pairs (oracle's fp32 / fp64 evaluations disagree):",
          [(int(k), f"{err32[k]:.2e}", f"{errtr[k]:.2e}", f"{err64[k]:.2e}") for k in undetermined])
    assert determined.sum() >= 256 - 11, msg
    assert err32[determined].max() < TOL_M and errtr[determined].max() < TOL_M, msg
    assert set(np.nonzero(err32 >= TOL_M)[0]) <= set(undetermined), msg
    assert (err32 >= TOL_M).sum() <= 6 and (errtr >= TOL_M).sum() <= 5, msg
    if len(undetermined):
        assert err32[undetermined].max() < 1e-2 and errtr[undetermined].max() < 2e-3, msg


def _vs_reference_import(name, S, D, o):
    """HIP against the G12 fixture `name` (the reference's OWN Python at this size, tools/gen_golden.py g12) and, beside it,
    the oracle run on this host.  -> (err vs fixture [B], message)."""
    from conftest import load_golden
    g = load_golden(name)
    B, N = int(g["num_pairs"]), int(g["max_points"])
    assert S.shape[:2] == (B, N)
    a = rp.default_args(max_points=N, icp_max_iterations=int(g["icp_max_iterations"]))
    init = utils_hist.estimate_init_pose(a, G(S), G(D)).cpu().numpy()
    assert np.array_equal(init, g["T_init"]), "initial poses differ from the reference-import run"
    T, iters = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    T = T.cpu().numpy()
    assert int(iters) == int(g["icp_iterations"]), (int(iters), int(g["icp_iterations"]))
    err = displacement(T, g["T_hist_icp"], S)
    und = np.nonzero(~o["determined"])[0]
    here = displacement(o["T32"], g["T_hist_icp"], S)
    exact = displacement(o["T64"], g["T_hist_icp"], S)
    out = np.nonzero(err >= TOL_M)[0]
    msg = "\n".join([
        f"{name}: HIP vs the reference-import run: {int((err < TOL_M).sum())}/{B} pairs within {TOL_M:g} m, max {err.max():.2e}, "
        f"outside: {[(int(k), f'{err[k]:.2e}') for k in out]}",
        f"  this host's fp32 oracle vs the reference-import run: {int((here == 0).sum())}/{B} pairs identical, max {here.max():.2e}",
        f"  the oracle's exact (fp64 Kabsch) evaluation vs the reference-import run: max {exact.max():.2e}, "
        f"outside {TOL_M:g}: {[(int(k), f'{exact[k]:.2e}') for k in np.nonzero(exact >= TOL_M)[0]]}",
        f"  pairs this host's oracle leaves undetermined, as the reference-import run decides them (HIP - run): "
        f"{[(int(k), f'{err[k]:.2e}') for k in und]}"])
    print(msg)
    # ev_* of the run on the run's own transforms, through the HIP match_eval (a sweep, no trajectory: tight)
    ev = utils_match.match_eval(a, G(S), G(D), G(g["T_hist_icp"]))
    for got, key in zip(ev, ("errors", "inliers", "ratios", "ious", "translations", "rotations")):
        np.testing.assert_allclose(got.cpu().numpy(), g["ev_" + key], atol=2e-5, rtol=1e-5, err_msg=key)
    return err, exact, msg


def test_config2_full_batch_vs_the_reference_import_run(config2):
    """G12: the whole config-2 batch against a run of the reference's own Python at this size (not the restatement).
    Initial poses bit for bit, the iteration count (47) equal, match_eval of the run's transforms within 2e-5.  Transforms:
    the HIP path's Kabsch moments are fp64, so it sits on the EXACT evaluation of each step (within 1e-5 m of the oracle's
    fp64-Kabsch run on every pair, previous test); the reference-import run is ONE fp32 evaluation of the same trajectory.
    Measured when the fixture was made: the run is further than 1e-4 m from the exact evaluation on 3 of 256 pairs (93: 1.4 mm,
    129: 1.2 mm, 127: 0.11 mm; pairs still sliding at the stop) -- the HIP path inherits exactly those, no others:
    a pair outside the bound must be one where the run itself is that far from the exact evaluation of its own formulas."""
    c = config2
    err, exact, msg = _vs_reference_import("g12_config2", c["S"], c["D"], c)
    assert (err >= TOL_M).sum() <= 5 and err.max() < 3e-3, msg
    assert set(np.nonzero(err >= TOL_M)[0]) <= set(np.nonzero(exact >= 0.5 * TOL_M)[0]), msg
    assert np.abs(err - exact).max() < 1e-5, msg    # HIP and the exact evaluation see the run from the same place


def test_config2_full_batch_fp32_reference_arithmetic(config2):
    """ICPFLOW_ARITH_FP32_REFERENCE (a study mode): the Kabsch step in the reference's own fp32 operation order, its
    sums as GPU tree reductions.  Same iteration count as the pairwise-order fp32 oracle and the exact evaluations
    (it is the ORDER of long fp32 accumulations, not fp32 itself, that keeps torch-CPU from converging); fp32
    rounding moves individual slowly-converging pairs by a few 1e-4 m, which is the scatter the reference's own
    CUDA run has against any other evaluation of itself."""
    c = config2
    with _lib.options(arith="fp32_reference"):
        T, iters = utils_match.hist_icp(c["args"], G(c["S"]), G(c["D"]), return_iterations=True)
    T = T.cpu().numpy()
    assert np.isfinite(T).all()
    err32 = displacement(T, c["T32"], c["S"])
    msg = (_report("HIP fp32-reference arithmetic vs fp32 oracle", err32, c["determined"])
           + f"\niterations HIP {int(iters)}, " + c["iters"])
    print(msg)
    assert int(iters) == c["auxtr"]["iterations"], msg
    assert (err32 < TOL_M).sum() >= 256 - 10 and err32[c["determined"]].max() < 1e-3, msg


@pytest.mark.parametrize("shape", ["config2_256x1024", "config4_shard_1024x2048", "ragged_600x1024", "ragged_400x3000",
                                   "teams_ragged_20x10000", "teams_12x6000"])
def test_adaptive_windows_change_nothing(shape):
    """Neighbour certificates, probes and reused moment sums (icp.hip): a query whose previous neighbour is
    provably still the nearest (or which provably has none inside the gate) is not searched, the few others of a wave
    are probed by rows of eight lanes, and a wave whose gated neighbours did not change keeps its sums.  Gate decisions,
    neighbours and sums are those of the plain window scan: transforms and iteration count are bit-identical to
    ICPFLOW_OPT_NO_ADAPTIVE_WINDOWS (the switch keeps its name from the first form of the per-query records)."""
    if shape == "config2_256x1024":
        S, D, _ = synthetic.make_batch(256, 1024, seed=0)
    elif shape == "config4_shard_1024x2048":
        S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    elif shape == "teams_ragged_20x10000":   # few large pairs: several workgroups per pair, records of each member's share
        S, D, _ = synthetic.make_batch(20, 10000, seed=7, ragged=True, n_min=500)
    elif shape == "teams_12x6000":
        S, D, _ = synthetic.make_batch(12, 6000, seed=9)
    elif shape == "ragged_600x1024":
        S, D, _ = synthetic.make_batch(600, 1024, seed=31, ragged=True, n_min=60)
    else:
        S, D, _ = synthetic.make_batch(400, 3000, seed=47, ragged=True, n_min=300)
    a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50)
    s, d = G(S), G(D)
    with _lib.options(no_adaptive_windows=True):
        T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it0) == int(it1) and int(it1) > 0
    assert torch.equal(T0, T1)
    ap = rp.default_args(max_points=S.shape[1], icp_max_iterations=50, icp_stop_mode="per_pair")
    with _lib.options(no_adaptive_windows=True):
        P0 = utils_match.hist_icp(ap, s, d)
    assert torch.equal(P0, utils_match.hist_icp(ap, s, d))


@pytest.mark.parametrize("shape", ["ragged_matched_128x10000", "ragged_independent_128x10000", "teams_ragged_20x10000", "teams_12x6000",
                                   "ragged_matched_40x4096"])
def test_shared_window_scans_change_nothing(shape):
    """Teams (icp.hip, round 5): a wave whose window holds 256 targets or more posts it in LDS, cut into parts; the waves of the
    member that are through with their own unit take parts and add their (minimum, runner-up, chunk | tie) to the owner's
    accumulator.  The merge gives what ONE scan over the window gives, whoever scanned which part: transforms and iteration
    counts are bit-identical to ICPFLOW_OPT_NO_SHARED_SCANS (the same kernel, every wave scanning alone), under the batch rule
    and per pair; and run-to-run (which wave takes which part is a race: the result must not depend on it)."""
    B, N, ragged, nmin, seed, cap = {"ragged_matched_128x10000": (128, 10000, "matched", 20, 0, 100),
                                      "ragged_independent_128x10000": (128, 10000, True, 20, 0, 100),
                                      "teams_ragged_20x10000": (20, 10000, True, 500, 7, 50),
                                      "teams_12x6000": (12, 6000, False, 20, 9, 50),
                                      "ragged_matched_40x4096": (40, 4096, "matched", 200, 3, 100)}[shape]
    S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=nmin)
    a = rp.default_args(max_points=N, icp_max_iterations=cap)
    s, d = G(S), G(D)
    with _lib.options(no_shared_scans=True):
        T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it0) == int(it1) and int(it1) > 0
    assert torch.isfinite(T1).all() and torch.equal(T0, T1)
    for _ in range(3):
        T2, it2 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert torch.equal(T1, T2) and int(it2) == int(it1)
    ap = rp.default_args(max_points=N, icp_max_iterations=cap, icp_stop_mode="per_pair")
    with _lib.options(no_shared_scans=True):
        P0 = utils_match.hist_icp(ap, s, d)
    assert torch.equal(P0, utils_match.hist_icp(ap, s, d))


@pytest.mark.parametrize("shape", ["config4_shard_1024x2048", "ragged_2500x700", "ragged_700x3000", "dense_1500x1024", "ragged_900x2048"])
def test_ticket_dispatch_changes_nothing(shape):
    """Batches larger than the GPU: the ICP launch is a grid as large as the GPU whose workgroups draw their further
    pairs from a ticket counter (icp.hip icp_kernel<..., PERSIST>) instead of one workgroup per pair dealt by the hardware
    dispatcher (round robin over the XCDs, in order: head-of-line blocking).  Which workgroup serves a pair changes
    nothing: transforms and iteration counts bit-identical to ICPFLOW_OPT_NO_PERSISTENT, under both stop rules."""
    if shape == "config4_shard_1024x2048":
        S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    elif shape == "ragged_2500x700":
        S, D, _ = synthetic.make_batch(2500, 700, seed=13, ragged=True, n_min=30)
    elif shape == "ragged_700x3000":
        S, D, _ = synthetic.make_batch(700, 3000, seed=17, ragged=True, n_min=200)
    elif shape == "ragged_900x2048":                      # (helpers on pairs of one to four passes, swapped roles)
        S, D, _ = synthetic.make_batch(900, 2048, seed=23, ragged=True, n_min=100)
    else:
        S, D, _ = synthetic.make_batch(1500, 1024, seed=19)
    s, d = G(S), G(D)
    for stop in ("reference", "per_pair"):
        a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50, icp_stop_mode=stop)
        with _lib.options(no_persistent=True):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and int(it1) > 0
        assert torch.equal(T0, T1)
        # Helpers (batches of up to four rounds of pairs with three passes or more, e.g. config 4's shard): a workgroup
        # that finds no ticket left takes whole passes of a pair another workgroup is still iterating on.  WHICH passes
        # come from helpers, and from which iteration on, depends on timing; the sums are kept per (pass, wave) and added
        # in that order whoever computed them, so nothing else does: without helpers, and run after run, the same bits.
        with _lib.options(no_helpers=True):
            T2, it2 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it2) == int(it1) and torch.equal(T2, T1)
        for _ in range(3):
            assert torch.equal(utils_match.hist_icp(a, s, d), T1)       # and from run to run
    with _lib.options(no_speculative=True):                             # one launch per iteration: the same sums again
        a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50)
        assert torch.equal(utils_match.hist_icp(a, s, d), utils_match.hist_icp(rp.default_args(max_points=S.shape[1], icp_max_iterations=50), s, d))


@pytest.mark.parametrize("shape", ["config4_shard_1024x2048", "ragged_900x2048", "dense_600x2048", "dense_1500x1500", "ragged_2000x1300", "cap_8_1024x2048",
                                   "masked_1024x1100"])
def test_two_icp_launches_change_nothing(shape):
    """Batches of a few rounds under the reference's batch-global stop (utils_icp_pytorch3d.py:153-213): the persistent grid of
    half-CU workgroups is DRAINED once the unfinished pairs fit one CU each -- they leave behind their current iteration, still
    moving -- and a second launch (icp.hip icp_split_kernel: the batch rule among the tallies, the list of those pairs) gives each
    a 1024-thread workgroup and resumes it at its own iteration from IcpState and its history rows.  Which iteration a pair is cut
    at depends on timing; nothing else does: the moment sums are added in the order of the 64-query units in either kernel, so
    transforms and iteration count are bit-identical to ONE launch (the default; ICPFLOW_OPT_TWO_LAUNCH switches the two launches on -- it is off because it does not pay), with and without helpers, and
    from run to run."""
    cap, mask = 50, None
    if shape == "config4_shard_1024x2048":
        S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    elif shape == "ragged_900x2048":                          # (the batch rule holds before the cap: 45 iterations)
        S, D, _ = synthetic.make_batch(900, 2048, seed=23, ragged=True, n_min=100)
    elif shape == "dense_600x2048":
        S, D, _ = synthetic.make_batch(600, 2048, seed=31)
    elif shape == "dense_1500x1500":
        S, D, _ = synthetic.make_batch(1500, 1500, seed=19)
    elif shape == "ragged_2000x1300":
        S, D, _ = synthetic.make_batch(2000, 1300, seed=13, ragged=True, n_min=30)
    elif shape == "cap_8_1024x2048":                          # a cap so low that pairs are cut in their first iterations
        S, D, _ = synthetic.make_batch(1024, 2048, seed=41)
        cap = 8
    else:                                                     # options.d_pair_active: a third of the pairs are not in the batch
        S, D, _ = synthetic.make_batch(1024, 1100, seed=37)
        mask = (np.arange(1024) % 3 != 1).astype(np.uint8)
        S[mask == 0] = np.array([1e8, 1e8, 1e8, 0], np.float32)
        D[mask == 0] = np.array([1e8, 1e8, 1e8, 0], np.float32)
    s, d = G(S), G(D)
    a = rp.default_args(max_points=S.shape[1], icp_max_iterations=cap)
    kw = dict(pair_active=G(mask)) if mask is not None else {}
    with _lib.options(**kw):                               # (the default: one launch)
        T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
    kw["two_launch"] = True
    with _lib.options(**kw):
        T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    keep = slice(None) if mask is None else torch.from_numpy(mask.astype(bool)).to(DEV)
    assert int(it0) == int(it1) and int(it1) > 0, (int(it0), int(it1))
    assert torch.equal(T0[keep], T1[keep])
    with _lib.options(no_helpers=True, **kw):             # (no helpers: one launch, the sums per (pass, wave) all the same)
        T2, it2 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it2) == int(it1) and torch.equal(T2[keep], T1[keep])
    with _lib.options(no_persistent=True, **kw):
        T3, it3 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it3) == int(it1) and torch.equal(T3[keep], T1[keep])
    for _ in range(4):
        with _lib.options(**kw):
            assert torch.equal(utils_match.hist_icp(a, s, d)[keep], T1[keep])
    # the ICP on its own (icpflow_icp) with its per-iteration records (t_history, :187): every iteration up to the stop, bit for bit
    if mask is None and shape in ("config4_shard_1024x2048", "ragged_900x2048"):
        from icp_flow_amd import utils_icp_pytorch3d as p3d
        one = p3d.iterative_closest_point(s, d, max_iterations=cap)
        with _lib.options(two_launch=True):
            two = p3d.iterative_closest_point(s, d, max_iterations=cap)
        assert torch.equal(one.RTs.R, two.RTs.R) and torch.equal(one.RTs.T, two.RTs.T) and torch.equal(one.rmse, two.rmse)
        assert len(one.t_history) == len(two.t_history) > 0
        for h1, h2 in zip(one.t_history, two.t_history):
            assert torch.equal(h1.R, h2.R) and torch.equal(h1.T, h2.T)


@pytest.mark.parametrize("shape", ["config4_shard_1024x2048", "dense_600x2048", "matched_128x4000", "matched_teams_48x10000", "independent_64x10000",
                                   "ragged_900x2048", "headings_along_the_axes_256x2048"])
def test_direction_sort_keys_change_only_the_order_of_the_sums(shape):
    """Round 6 (csrc/sortdir.hpp): the clouds of a pair are sorted by the key that spreads the fixed cloud best -- an axis or one of six
    horizontal directions -- instead of always along its longest axis (ICPFLOW_OPT_NO_DIR_KEYS).  Every search stays exact
    (|u . (q - t)| <= |q - t| for any unit u; a computed key gives its rounding away first): same neighbours, same gate decisions,
    same iteration count.  What changes is the ORDER in which a pair's queries are visited, i.e. the order of the fp64 moment sums:
    a transform may differ in its last bit (measured: 1 pair in 8192, 1.5e-13 m).  So: iteration counts equal, every point within
    1e-9 m, nearly all transforms bit-identical -- and the scoring's picks (integer bins, exact minima) equal bit for bit."""
    if shape == "config4_shard_1024x2048":
        S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    elif shape == "dense_600x2048":
        S, D, _ = synthetic.make_batch(600, 2048, seed=31)
    elif shape == "matched_128x4000":
        S, D, _ = synthetic.make_batch(128, 4000, seed=3, ragged="matched", n_min=200)
    elif shape == "matched_teams_48x10000":                  # long clouds: the chunked sorts (sort.hip), teams with shared scans
        S, D, _ = synthetic.make_batch(48, 10000, seed=5, ragged="matched", n_min=400)
    elif shape == "independent_64x10000":
        S, D, _ = synthetic.make_batch(64, 10000, seed=7, ragged=True, n_min=50)
    elif shape == "ragged_900x2048":
        S, D, _ = synthetic.make_batch(900, 2048, seed=23, ragged=True, n_min=100)
    else:
        # boxes whose heading is exactly along x or y: a face across either axis -- the case the direction keys exist for
        S, D, _ = synthetic.make_batch(256, 2048, seed=11)
        for k in range(256):
            r = np.random.default_rng(900 + k)
            ext = np.array([r.uniform(2.5, 5.0), r.uniform(1.2, 2.2), r.uniform(1.0, 2.0)])
            pts = synthetic._shell_points(r, ext, 2048)
            if k % 2:
                pts = pts[:, [1, 0, 2]]
            c = np.array([r.uniform(-30, 30), r.uniform(-30, 30), 0.8])
            t = np.array([r.uniform(-1, 1), r.uniform(-1, 1), 0.0])
            S[k, :, :3] = (pts + c).astype(np.float32); S[k, :, 3] = 1.0
            D[k, :, :3] = (pts + c + t + r.normal(0, 0.01, pts.shape)).astype(np.float32); D[k, :, 3] = 1.0
    s, d = G(S), G(D)
    a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50)
    with _lib.options(no_dir_keys=True):
        T0, ev0, it0 = utils_match.hist_icp_eval(a, s, d, return_iterations=True)
        P0 = utils_hist.estimate_init_pose(a, s, d)
    T1, ev1, it1 = utils_match.hist_icp_eval(a, s, d, return_iterations=True)
    P1 = utils_hist.estimate_init_pose(a, s, d)
    assert torch.equal(P0, P1)                                   # the initial poses: exact minima, integer bins
    assert int(it0) == int(it1) and int(it1) > 0
    dis = displacement(T0.cpu().numpy(), T1.cpu().numpy(), S)
    same = (T0 == T1).flatten(1).all(1).float().mean().item()
    assert dis.max() < 1e-9, (shape, float(dis.max()))
    assert same >= 0.99, (shape, same)
    for e0, e1 in zip(ev0, ev1):
        assert torch.allclose(e0, e1, rtol=1e-6, atol=1e-7)
    for _ in range(2):                                          # and from run to run the same bits
        assert torch.equal(utils_match.hist_icp(a, s, d), T1)


def test_per_pair_stop_with_a_long_iteration_cap_keeps_the_helper_protocol_sound():
    """the helper hand-off words carry the iteration epoch in 8 bits; ICPFLOW_STOP_PER_PAIR runs ONE persistent
    launch of up to 1024 iterations, so a cap beyond 253 must not be served with helpers (launch_icp switches them off:
    kHelpMaxEpoch).  A helper-eligible shape (more pairs than workgroup slots, several passes per pair) at
    max_iterations = 512: finite, no abandoned team, and bit-identical to ICPFLOW_OPT_NO_HELPERS; the same at a cap of
    200, where helpers do run."""
    S, D, _ = synthetic.make_batch(900, 2048, seed=23, ragged=True, n_min=100)
    s, d = G(S), G(D)
    for cap in (512, 200):
        a = rp.default_args(max_points=S.shape[1], icp_max_iterations=cap, icp_stop_mode="per_pair")
        T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it1) > 0 and torch.isfinite(T1).all()
        with _lib.options(no_helpers=True):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and torch.equal(T0, T1)


@pytest.mark.parametrize("shape", ["config2_256x1024", "config4_shard_1024x2048", "ragged_600x1024", "ragged_90x2048", "small_against_long_48x10000"])
def test_scoring_variants_change_nothing(shape):
    """The six candidate translations are scored by sorted sweeps with branch and bound when hist_icp has the clouds
    sorted anyway (nn.hip launch_sweep_score_pruned: blocks and waves leave a scan once its running LOWER bound rules
    the candidate out), by the pruned all-pairs scans otherwise.  Pruning never changes the pick, so the registrations
    are bit-identical to the pruned all-pairs scoring (ICPFLOW_OPT_NO_SCORE_SWEEP) and to every scan run to its end
    (ICPFLOW_OPT_NO_SCORE_PRUNE) -- and the same from run to run, although WHICH scans get pruned depends on timing."""
    if shape == "config2_256x1024":
        S, D, _ = synthetic.make_batch(256, 1024, seed=0)
    elif shape == "config4_shard_1024x2048":
        S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    elif shape == "ragged_600x1024":
        S, D, _ = synthetic.make_batch(600, 1024, seed=31, ragged=True, n_min=60)
    elif shape == "small_against_long_48x10000":
        # (round 5) clusters of 20 ... 10^4 points, the two clouds of a pair independent: a 40-point cloud against a 7000-point one --
        # sweep blocks of ONE wave share their window over the block's four waves (512 <= targets < 2048), one-block clouds are
        # scanned by eight blocks that leave partial minima in global memory (targets >= 2048): the same sums bit for bit, hence the
        # same picks, poses and metrics as the all-pairs scans
        S, D, _ = synthetic.make_batch(48, 10000, seed=0, ragged=True, n_min=20)
    else:
        S, D, _ = synthetic.make_batch(90, 2048, seed=5, ragged=True, n_min=40)
    a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50)
    s, d = G(S), G(D)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    if shape == "small_against_long_48x10000":
        # ... and the roll-back check and match_eval sweeps against their all-pairs scans
        with _lib.options(no_check_sweep=True):
            Tc = utils_match.hist_icp(a, s, d)
        assert torch.equal(Tc, T1)
        ev1 = utils_match.match_eval(a, s, d, T1)
        with _lib.options(no_eval_sweep=True):
            ev0 = utils_match.match_eval(a, s, d, T1)
        for x, y in zip(ev1, ev0):
            assert torch.equal(x, y)
    with _lib.options(no_score_sweep=True):
        T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
    with _lib.options(no_score_prune=True):
        T2, it2 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it0) == int(it1) == int(it2) and int(it1) > 0
    assert torch.equal(T0, T1) and torch.equal(T2, T1)
    # (header 0.2.11) the roll-back check takes its sum under the initial pose from the scoring's forward scan of the picked
    # candidate -- the same scan of the same points (pose.hip score_pick_kernel, nn.hip sweep_scan_kernel) -- unless that scan
    # was pruned; with ICPFLOW_OPT_NO_CHECK_REUSE the check scans under both poses as before: same poses bit for bit
    with _lib.options(no_check_reuse=True):
        T3, it3 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it3) == int(it1) and torch.equal(T3, T1)
    with _lib.options(no_check_reuse=True, no_score_prune=True):
        assert torch.equal(utils_match.hist_icp(a, s, d), T1)
    # (header 0.2.12) the pruned sweeps first bound a scan's WHOLE sum from below by the other cloud's dilated occupancy grid (queries
    # in cells with no target in the 27-neighbourhood x 0.98 h: nn.hip occ_build_kernel) and leave before evaluating a target where that
    # already rules the candidate out; without the grids (ICPFLOW_OPT_NO_SCORE_PREBOUND) the same picks, hence the same poses
    with _lib.options(no_score_prebound=True):
        T4, it4 = utils_match.hist_icp(a, s, d, return_iterations=True)
    assert int(it4) == int(it1) and torch.equal(T4, T1)
    init1 = utils_hist.estimate_init_pose(a, s, d)
    with _lib.options(no_score_prebound=True):
        assert torch.equal(utils_hist.estimate_init_pose(a, s, d), init1)
    for _ in range(3):
        assert torch.equal(utils_match.hist_icp(a, s, d), T1)


@pytest.mark.parametrize("offset,scale", [(0.0, 1.0), (150.0, 1.0), (900.0, 1.0), (6000.0, 1.0), (40.0, 12.0), (0.0, 0.02)])
def test_occupancy_prebound_never_changes_a_pick(offset, scale):
    """(header 0.2.12) The pre-bound of the pruned scoring sweeps counts the queries whose cell of the other cloud's dilated occupancy
    grid is empty: such a query has no target within 0.98 h, whatever the rounding of the cell arithmetic.  Clouds far from the origin
    (cell arithmetic on coordinates of hundreds of metres; beyond ~2.6 km the build refuses the grid), clusters tens of metres across
    (the cell edge grows until the grid fits) and clusters of centimetres (a handful of cells): the initial poses equal the oracle's
    and those of the sweeps without the grids; the registrations are the same bit for bit."""
    N = 1100
    S, D, _ = synthetic.make_batch(48, N, seed=77, ragged=True, n_min=3)
    for b, sd in ((7, 2), (19, 3)):
        S[b], D[b] = _backward_winner_pair(N, sd)
    for A in (S, D):
        v = A[:, :, 3] > 0
        A[:, :, :3][v] = A[:, :, :3][v] * np.float32(scale) + np.float32(offset)
    a = rp.default_args(max_points=N, icp_max_iterations=30)
    s, d = G(S), G(D)
    init = utils_hist.estimate_init_pose(a, s, d)
    with _lib.options(no_score_prebound=True):
        init0 = utils_hist.estimate_init_pose(a, s, d)
    assert torch.equal(init, init0)
    want = rp.estimate_init_pose(a, C(S), C(D)).numpy()
    if scale >= 1.0:
        assert np.array_equal(init.cpu().numpy(), want)
    else:   # centimetre clusters: the whole vote sits in a bin or two and the other "peaks" are ties among empty bins, whose order is
        # torch.topk's (COVERAGE.md, named deviations) -- reported, not asserted; the variants below still agree bit for bit
        print("pairs whose pick differs from the oracle's (tied peaks):", int((np.abs(init.cpu().numpy() - want).max((1, 2)) > 0).sum()))
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    for opts in ({"no_score_prebound": True}, {"no_score_prune": True}, {"no_score_prebound": True, "no_check_reuse": True}):
        with _lib.options(**opts):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and torch.equal(T0, T1), opts


@pytest.mark.parametrize("offset", [0.0, 6000.0])
def test_pruned_scoring_in_two_launches_changes_nothing(offset):
    """(header 0.2.12) From 65 536 (scan, query block) jobs on, the pruned scoring launch comes in two: a deciding launch of one block
    per scan and sweep_list_kernel, whose eight workgroups per CU stride over the jobs of the scans that go on.  A ragged batch of
    config 4's size: with the occupancy grids (a hundred-odd scans go on) and 6 km from the origin, where the build refuses the grids
    and EVERY scan goes on -- ten thousand listed scans, forty jobs per workgroup of the second launch: the registrations are those
    of the one plain launch (ICPFLOW_OPT_NO_SCORE_PREBOUND) and of every scan run to its end."""
    B, N = 1024, 2048
    S, D, _ = synthetic.make_batch(B, N, seed=91, ragged=True, n_min=30)
    for A in (S, D):
        v = A[:, :, 3] > 0
        A[:, :, :3][v] = A[:, :, :3][v] + np.float32(offset)
    a = rp.default_args(max_points=N, icp_max_iterations=20)
    s, d = G(S), G(D)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    for opts in ({"no_score_prebound": True}, {"no_score_prune": True}):
        with _lib.options(**opts):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and torch.equal(T0, T1), opts
    assert torch.equal(utils_match.hist_icp(a, s, d), T1)


def _backward_winner_pair(N, seed):
    """src role: a 100-point patch P and a tight 300-point clump K three metres above it (out of the vote's z range of each other);
    dst role: ten jittered copies of P moved by t1 = (-0.3, 0.4, 0) and 20 points of K moved by t0 = (0.4, 0.4, 0).  The vote's
    highest peak is t0 (300 x 20 votes in one bin), seven bins from t1 (outside the 11-bin suppression window of
    utils_hist.py:21-24).  t0's forward mean is moderate (the clump on its copies, the patch 0.7 m beside its own); t1 has the
    LARGER forward mean (the clump 0.7 m off: 0.52 against 0.08) and by far the smallest backward mean (1000 of the 1020 dst
    points sit on the patch: 0.018): t1 wins through its backward scan while its forward scan is cut off by candidate 0's bound."""
    rng = np.random.default_rng(seed)
    P = np.stack([rng.uniform(0, 1, 100), rng.uniform(0, 1, 100), rng.uniform(-0.01, 0.01, 100)], 1)
    K = np.array([0.5, 0.5, 3.0]) + rng.uniform(-0.01, 0.01, (300, 3))
    t0, t1 = np.array([0.4, 0.4, 0.0]), np.array([-0.3, 0.4, 0.0])
    A = np.concatenate([P, K])
    Cc = np.concatenate([np.repeat(P, 10, 0) + t1 + rng.normal(0, 0.003, (1000, 3)), K[:20] + t0])
    S = np.full((N, 4), 1e8, np.float32); S[:, 3] = 0
    D = S.copy()
    S[:len(A), :3] = A + 10.0; S[:len(A), 3] = 1
    D[:len(Cc), :3] = Cc + 10.0; D[:len(Cc), 3] = 1
    return S, D


def test_check_scans_for_itself_where_the_picked_candidates_forward_scan_was_pruned():
    """(header 0.2.11) The roll-back check takes its sum under the initial pose from the scoring -- the forward total of the picked
    candidate -- unless that scan was pruned: a candidate can win through its BACKWARD mean while its forward scan exceeded
    candidate 0's bound and reports +inf.  Two such pairs (built for it; tools/dbg/check_reuse_stats.py counts them in a
    -DICPFLOW_REUSE_STATS build) among ordinary ones: the pick is the backward winner, as in the oracle, and the registrations are
    bit for bit those of the check that scans under both poses (ICPFLOW_OPT_NO_CHECK_REUSE), of every scoring scan run to its end
    (where the total is finite and IS taken over) and of the all-pairs check."""
    N = 1100
    S, D, _ = synthetic.make_batch(8, N, seed=41, ragged=True, n_min=300)
    for b, seed in ((2, 0), (5, 1)):
        S[b], D[b] = _backward_winner_pair(N, seed)
    a = rp.default_args(max_points=N, icp_max_iterations=50)
    s, d = G(S), G(D)
    init = utils_hist.estimate_init_pose(a, s, d).cpu().numpy()
    want_init = rp.estimate_init_pose(a, C(S), C(D)).numpy()
    assert np.array_equal(init, want_init)
    for b in (2, 5):
        assert np.allclose(init[b, :3, 3], [-0.3, 0.4, 0.0], atol=1e-6)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    for opts in ({"no_check_reuse": True}, {"no_score_prune": True}, {"no_check_sweep": True}, {"no_check_reuse": True, "no_score_prune": True},
                 {"no_score_prebound": True}):
        with _lib.options(**opts):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and torch.equal(T0, T1), opts
    want = rp.hist_icp(a, C(S[[2, 5]]), C(D[[2, 5]]), max_iterations=int(it1), kabsch_dtype=torch.float64).numpy()
    err = displacement(T1.cpu().numpy()[[2, 5]], want, S[[2, 5]])
    print("backward winners against the oracle (same iteration count): %.2e m" % err.max())
    assert err.max() < 1e-4


@pytest.mark.parametrize("shape", ["config2_256x1024", "ragged_90x2048", "long_clouds_6x6000"])
def test_launch_plumbing_variants_change_nothing(shape):
    """hist_icp's launch plumbing: where one workgroup sorts a cloud (N <= 4096) the vote's sort counts the valid rows,
    writes lengths and swap flag and clears the scratch, and the axis sort on the side stream counts for itself;
    count_pair_kernel does it otherwise.  Without the side stream (ICPFLOW_OPT_NO_SIDE_STREAM), with the all-pairs vote
    (ICPFLOW_OPT_NO_SORTED_VOTE: count_pair, identical bins) and with both, the registrations are the same bit for bit."""
    if shape == "config2_256x1024":
        S, D, _ = synthetic.make_batch(256, 1024, seed=0)
    elif shape == "ragged_90x2048":
        S, D, _ = synthetic.make_batch(90, 2048, seed=5, ragged=True, n_min=40)
    else:
        S, D, _ = synthetic.make_batch(6, 6000, seed=9, ragged=True, n_min=3000)
    a = rp.default_args(max_points=S.shape[1], icp_max_iterations=50)
    s, d = G(S), G(D)
    T1, it1 = utils_match.hist_icp(a, s, d, return_iterations=True)
    for opts in ({"no_side_stream": True}, {"no_sorted_vote": True}, {"no_side_stream": True, "no_sorted_vote": True}):
        with _lib.options(**opts):
            T0, it0 = utils_match.hist_icp(a, s, d, return_iterations=True)
        assert int(it0) == int(it1) and torch.equal(T0, T1), opts


def _adversarial_batch():
    """Cluster pairs built to stress the certificates' bounds: exact distance ties (lattice points, duplicated targets),
    coordinates of a few km (fp32 ulp 2.4e-4 m .. 4.9e-4 m: the rounding of the window bounds and of the moved points is
    of the size of the margins), tiny clouds, clouds without a single neighbour inside the gate, and plain pairs."""
    rng = np.random.default_rng(2024)
    B, N = 96, 700
    S = np.full((B, N, 4), 1e8, np.float32)
    D = np.full((B, N, 4), 1e8, np.float32)
    S[:, :, 3] = D[:, :, 3] = 0.0
    for b in range(B):
        kind = b % 6
        ns = int(rng.integers(3, N + 1)) if kind == 4 else int(rng.integers(200, N + 1))
        nd = int(rng.integers(3, N + 1)) if kind == 4 else int(rng.integers(200, N + 1))
        ext = np.array([rng.uniform(1.0, 5.0), rng.uniform(0.8, 2.0), rng.uniform(0.5, 1.8)])
        centre = rng.uniform(-40, 40, 3)
        if kind == 1:                                         # lattice: many exactly equal distances
            g = 0.0625
            src = np.floor(rng.uniform(-0.5, 0.5, (ns, 3)) * ext / g) * g
            dst = np.floor(rng.uniform(-0.5, 0.5, (nd, 3)) * ext / g) * g + np.array([g / 2, 0.0, 0.0])
            centre = np.round(centre / g) * g
        else:
            src = rng.uniform(-0.5, 0.5, (ns, 3)) * ext
            dst = rng.uniform(-0.5, 0.5, (nd, 3)) * ext
            if kind in (0, 2, 3, 4):                          # overlapping surfaces, small motion
                m = min(ns, nd)
                dst[:m] = src[:m] + rng.normal(0, 0.01, (m, 3))
            dst = dst + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 0.02])
        if kind == 2:                                         # duplicated targets: the first-index rule decides
            dup = rng.integers(0, nd, nd // 3)
            dst[rng.integers(0, nd, nd // 3)] = dst[dup]
        if kind == 3:                                         # kilometres from the origin
            centre = centre + np.array([4000.0, -7000.0, 0.0])
        if kind == 5:                                         # nothing inside the gate
            dst = dst + np.array([3.0, 0.0, 0.0])
        S[b, :ns, :3] = src + centre; S[b, :ns, 3] = 1.0
        D[b, :nd, :3] = dst + centre; D[b, :nd, 3] = 1.0
    return S, D


def test_certificates_on_adversarial_clouds_change_nothing():
    """The neighbour certificates and probes (icp.hip) against the plain window scan on clouds built to hit their
    bounds: ties, duplicates, km-sized coordinates, tiny clouds, empty gates.  Bit-identical transforms, iteration
    counts and per-pair results, straight ICP (no vote in front: large first steps) and hist_icp."""
    from icp_flow_amd import utils_icp_pytorch3d
    S, D = _adversarial_batch()
    s, d = G(S), G(D)
    for cap, stop in ((60, "reference"), (25, "per_pair")):
        kw = dict(max_iterations=cap, stop_mode=stop)
        with _lib.options(no_adaptive_windows=True):
            a0 = utils_icp_pytorch3d.iterative_closest_point(s, d, **kw)
            R0, T0, e0, x0 = a0.RTs.R.clone(), a0.RTs.T.clone(), a0.rmse.clone(), a0.Xt.clone()
            n0 = a0.converged.iterations
        a1 = utils_icp_pytorch3d.iterative_closest_point(s, d, **kw)
        assert a1.converged.iterations == n0
        assert torch.equal(a1.RTs.R, R0) and torch.equal(a1.RTs.T, T0)
        assert torch.equal(torch.nan_to_num(a1.rmse, nan=-1.0), torch.nan_to_num(e0, nan=-1.0))
        assert torch.equal(torch.nan_to_num(a1.Xt, nan=-1.0), torch.nan_to_num(x0, nan=-1.0))
    for stop in ("reference", "per_pair"):
        a = rp.default_args(max_points=S.shape[1], icp_max_iterations=40, icp_stop_mode=stop)
        with _lib.options(no_adaptive_windows=True):
            T0 = utils_match.hist_icp(a, s, d)
        assert torch.equal(T0, utils_match.hist_icp(a, s, d))
    # more than 128 iterations: one launch per iteration (the records do not outlive a launch: every launch scans)
    with _lib.options(no_adaptive_windows=True):
        a0 = utils_icp_pytorch3d.iterative_closest_point(s[:24], d[:24], max_iterations=140)
        R0, n0 = a0.RTs.R.clone(), a0.converged.iterations
    a1 = utils_icp_pytorch3d.iterative_closest_point(s[:24], d[:24], max_iterations=140)
    assert a1.converged.iterations == n0 and torch.equal(a1.RTs.R, R0)


def _boxes_with_faces(B, N, seed):
    """Axis-aligned boxes sampled on their surface with a THIRD of the points on the two faces perpendicular to the long
    axis: hundreds of targets with exactly one coordinate along the ICP's sort axis (the case the long probes exist for,
    icp.hip kProbeStepsLong).  Small yaw and translation, independently resampled destination."""
    rng = np.random.default_rng(seed)
    S = np.full((B, N, 4), 1e8, np.float32); D = np.full((B, N, 4), 1e8, np.float32)
    S[:, :, 3] = D[:, :, 3] = 0.0
    for b in range(B):
        ext = np.array([rng.uniform(3.0, 5.0), rng.uniform(1.6, 2.2), rng.uniform(1.4, 2.0)])
        centre = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), 0.8])
        def sample(n):
            p = rng.uniform(-0.5, 0.5, (n, 3)) * ext
            face = rng.random(n) < 1.0 / 3.0
            p[face, 0] = np.where(rng.random(face.sum()) < 0.5, -0.5, 0.5) * ext[0]        # the two end faces: exact coordinate
            side = ~face
            ax = rng.integers(1, 3, side.sum())
            sgn = np.where(rng.random(side.sum()) < 0.5, -0.5, 0.5)
            q = p[side]; q[np.arange(len(q)), ax] = sgn * ext[ax]; p[side] = q
            return p
        ns, nd = int(rng.integers(N // 2, N + 1)), int(rng.integers(N // 2, N + 1))
        yaw = np.deg2rad(rng.uniform(-2, 2)); c, s_ = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
        t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0])
        S[b, :ns, :3] = sample(ns) + centre; S[b, :ns, 3] = 1.0
        D[b, :nd, :3] = sample(nd) @ R.T + centre + t + rng.normal(0, 0.01, (nd, 3)); D[b, :nd, 3] = 1.0
    return S, D


@pytest.mark.parametrize("shape", ["two_per_cu_600x2048", "one_per_cu_200x3000", "teams_20x6000"])
def test_long_probes_on_faces_perpendicular_to_the_sort_axis_change_nothing(shape):
    """Clouds of several passes whose end faces hold hundreds of points with ONE coordinate along the sort axis: the
    probes there need more than ten blocks to prove anything (kProbeStepsLong = 24).  Against the plain window scan
    (`no_adaptive_windows`): transforms and iteration counts bit-identical, single workgroups, two per CU and teams."""
    B, N = {"two_per_cu_600x2048": (600, 2048), "one_per_cu_200x3000": (200, 3000), "teams_20x6000": (20, 6000)}[shape]
    S, D = _boxes_with_faces(B, N, seed=77)
    a = rp.default_args(max_points=N, icp_max_iterations=40)
    with _lib.options(no_adaptive_windows=True):
        T0, it0 = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    T1, it1 = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    assert int(it0) == int(it1) and torch.equal(T0, T1)
    assert torch.isfinite(T1).all()


# ------------------------------------------------------------------------------------------ fused vote bins
def _wide_pair(n, seed):
    """A wall: 24 m x 0.4 m x 2.6 m, n points -- wider than the 8 m above which the vote sorts by the composite
    key (votekey.hpp) -- and the same wall 0.7 m further along, re-sampled."""
    r = np.random.default_rng(seed)
    def wall(m):
        p = np.stack([r.uniform(0, 24.0, m), r.uniform(0, 0.4, m), r.uniform(0, 2.6, m)], 1)
        return p + np.array([12.0, -30.0, 0.2])
    def pad(p):
        out = np.zeros((n, 4), np.float32)
        out[:, :3] = p
        out[:, 3] = 1.0
        return out
    return pad(wall(n)), pad(wall(n) + np.array([0.7, 0.05, 0.0]))


def _fused_cases():
    S2, D2, _ = synthetic.make_batch(256, 1024, seed=0)
    Sr, Dr, _ = synthetic.make_batch(24, 700, seed=11, ragged=True)
    w = [_wide_pair(3000, s) for s in (1, 2)]
    Sw, Dw = np.stack([a for a, _ in w]), np.stack([b for _, b in w])
    Sc, Dc, _ = synthetic.make_batch(3, 6000, seed=5)
    Sc[1, 4000:] = (1e8, 1e8, 1e8, 0.0)                      # one shorter cloud in the chunk-sorted batch
    Sg, Dg, _ = synthetic.make_batch(6, 900, seed=21, ragged=True)
    Sv, Dv, _ = synthetic.make_batch(12, 5000, seed=31, ragged=True, n_min=20)   # the vote's work list: very unequal pairs,
    Sv[3, :, 3] = 0.0                                                            # one without a single src row
    Sv[3, :, :3] = 1e8
    return {"ragged_wide_12x5000_work_list_tf2": (Sv, Dv, 2.0), "config2_256x1024_tf2": (S2, D2, 2.0), "ragged_24x700_tf2": (Sr, Dr, 2.0),
            "wide_cluster_composite_key_tf3p34": (Sw, Dw, 3.34), "chunked_sort_N6000_tf2": (Sc, Dc, 2.0),
            "global_atomics_269_bins_tf13p36": (Sg, Dg, 13.36)}


@pytest.mark.parametrize("case", ["config2_256x1024_tf2", "ragged_24x700_tf2", "wide_cluster_composite_key_tf3p34",
                                  "chunked_sort_N6000_tf2", "global_atomics_269_bins_tf13p36",
                                  "ragged_wide_12x5000_work_list_tf2"])
@pytest.mark.parametrize("entry", ["estimate_init_pose", "hist_icp"])
def test_fused_vote_bins_bit_exact(case, entry):
    """The vote that the registration path actually runs (zsort_kernel + hist_vote_sorted_kernel, composite key for
    wide clusters, chunked sort above 4096 points, global atomics for histograms beyond the LDS): its uint32 bins,
    exported through icpflow_options_t.d_vote_bins_u32, equal the oracle's vote (hist_cuda_core.cuh:40-60) bin for
    bin.  Through hist_icp the clouds of swapped pairs vote in exchanged roles (utils_match.py:139-146)."""
    S, D, tf = _fused_cases()[case]
    _all_host_threads()
    a = rp.default_args(max_points=S.shape[1], translation_frame=tf, icp_max_iterations=2)
    ex, ey, ez = rp.bin_edges(a)
    L = len(ex) * len(ey) * len(ez)
    bins = torch.full((len(S), L), -1, dtype=torch.int32, device=DEV)      # uint32 on the device, every bin overwritten
    with _lib.options(vote_bins=bins):
        if entry == "estimate_init_pose":
            utils_hist.estimate_init_pose(a, G(S), G(D))
            src, dst = C(S), C(D)
        else:
            utils_match.hist_icp(a, G(S), G(D))
            n1, n2 = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
            sw = n1 > n2                                                            # utils_match.py:142
            src, dst = C(S).clone(), C(D).clone()
            src[sw], dst[sw] = C(D)[sw], C(S)[sw]
    want = rp.hist(dst, src, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex), len(ey), len(ez))
    got = bins.cpu().numpy().view(np.uint32).astype(np.int64)
    want = want.numpy().reshape(len(S), L).astype(np.int64)
    assert want.sum() > 0
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} bins differ"
    # ... and the same bins from the grid over the padded widths (ICPFLOW_OPT_NO_VOTE_LIST: no work list where one is used)
    plain = torch.full((len(S), L), -1, dtype=torch.int32, device=DEV)
    with _lib.options(vote_bins=plain, no_vote_list=True):
        (utils_hist.estimate_init_pose if entry == "estimate_init_pose" else utils_match.hist_icp)(a, G(S), G(D))
    assert torch.equal(plain, bins)


# ------------------------------------------------------------------------------------------ config 4 (per-GPU shape)
@pytest.fixture(scope="module")
def config4_shard():
    """Rank 0's shard of BASELINE config 4 (8192 x 2048 over 8 GPUs): pairs 0..1023, 2048 points each."""
    S, D, Tt = synthetic.make_batch(1024, 2048, seed=0)
    return S, D, Tt


def test_config4_shard_properties(config4_shard):
    """1024 pairs x 2048 points in one call: finite, motion recovered on the shared-sample pairs, running the
    shard in two halves changes nothing but what the batch-global stop couples (iteration counts), and the result
    is reproducible bit for bit."""
    S, D, Tt = config4_shard
    a = rp.default_args(max_points=2048, icp_max_iterations=50)
    s, d = G(S), G(D)
    T, iters = utils_match.hist_icp(a, s, d, return_iterations=True)
    T2 = utils_match.hist_icp(a, s, d)
    assert torch.equal(T, T2)                                           # run-to-run bit-reproducible
    T = T.cpu().numpy()
    assert np.isfinite(T).all() and 1 <= int(iters) <= 50
    err = displacement(T, Tt, S)
    assert np.median(err[0::2]) < 0.005 and err[0::2].max() < 0.03     # shared-sample pairs: the true motion
    # the initial poses do not depend on the batch: the first 64 pairs alone give the same ones
    i_full = utils_hist.estimate_init_pose(a, s, d)
    i_part = utils_hist.estimate_init_pose(a, s[:64].contiguous(), d[:64].contiguous())
    assert torch.equal(i_full[:64], i_part)
    # per-pair stop: every pair on its own; same poses wherever the batch rule had converged pairs waiting
    ap = rp.default_args(max_points=2048, icp_max_iterations=50, icp_stop_mode="per_pair")
    Tp = utils_match.hist_icp(ap, s, d).cpu().numpy()
    assert np.percentile(displacement(T, Tp, S), 90) < TOL_M


@pytest.fixture(scope="module")
def config4_sample_oracle(config4_shard):
    S, D, _ = config4_shard
    return _oracle_three(rp.default_args(max_points=2048, icp_max_iterations=50), S[:64], D[:64], 50)


def test_config4_shape_64_pair_batch_vs_oracle(config4_shard, config4_sample_oracle):
    S, D, _ = config4_shard
    S, D = S[:64], D[:64]
    a = rp.default_args(max_points=2048, icp_max_iterations=50)
    o = config4_sample_oracle
    init = utils_hist.estimate_init_pose(a, G(S), G(D)).cpu().numpy()
    assert np.array_equal(init, o["aux32"]["init"].numpy())
    T, iters = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    T = T.cpu().numpy()
    determined = o["determined"]
    err32 = displacement(T, o["T32"], S)
    err64 = displacement(T, o["T64"], S)
    msg = (_report("HIP vs fp32 oracle (64 x 2048)", err32, determined) + "\n" + _report("HIP vs fp64-Kabsch oracle", err64, determined)
           + f"\niterations HIP {int(iters)}, " + o["iters"])
    print(msg)
    assert int(iters) == o["aux64"]["iterations"] == o["auxtr"]["iterations"], msg
    assert err64.max() < 1e-5, msg               # every pair, no mask, against the exact evaluation of the oracle
    # the reference's fp32 arithmetic: every pair it pins (every pair runs into the cap of 50 here: more of them are still
    # moving, and fp32 summation order decides where those end up -- a report on the oracle's rounding, see config 2)
    # (measured on the MI355X box's host, 32 threads: 56 determined pairs of 64, three pairs -- 45, 7, 57 -- beyond 1e-4 m of
    # the torch-order fp32 evaluation, 0.6 mm at most: counts + 2, and a ceiling on the excused pairs)
    undetermined = np.nonzero(~determined)[0]
    print("excused pairs:", [(int(k), f"{err32[k]:.2e}", f"{err64[k]:.2e}") for k in undetermined])
    assert determined.sum() >= 64 - 10, msg
    assert err32[determined].max() < TOL_M, msg
    assert set(np.nonzero(err32 >= TOL_M)[0]) <= set(undetermined), msg
    assert (err32 >= TOL_M).sum() <= 5, msg
    if len(undetermined):
        assert err32[undetermined].max() < 2e-3, msg


def test_config4_shape_64_pair_batch_vs_the_reference_import_run(config4_shard, config4_sample_oracle):
    """G12: config 4's 64-pair sample against the reference's own Python run (50 iterations, not converged: every pair is
    cut off at the cap while some still move).  Measured when the fixture was made: the run is > 1e-4 m from the exact
    evaluation of its own formulas on 4 of 64 pairs (7: 0.30 mm, 61: 0.19 mm, 57: 0.15 mm, 53: 0.14 mm)."""
    S, D, _ = config4_shard
    S, D = S[:64], D[:64]
    err, exact, msg = _vs_reference_import("g12_config4_sample", S, D, config4_sample_oracle)
    assert (err >= TOL_M).sum() <= 6 and err.max() < 1e-3, msg
    assert set(np.nonzero(err >= TOL_M)[0]) <= set(np.nonzero(exact >= 0.5 * TOL_M)[0]), msg
    assert np.abs(err - exact).max() < 1e-5, msg


# ------------------------------------------------------------------------------------------ p = len (no clamp)
@pytest.mark.parametrize("tf", [2.0, 13.36])
def test_vote_whose_quotient_rounds_to_one_lands_in_the_next_pair(tf):
    """hist_cuda_core.cuh:52-58 does not clamp: a difference one float below max gives (v - min) / (max - min) == 1.0,
    p_x = len_x, and the flat bin index runs into the NEXT pair's bins (one [B, L] allocation, hist_cuda.cu:59); past
    the last pair the reference writes out of bounds (undefined; dropped here, never written).  Both vote kernels --
    the public hist() and the fused z-sorted vote, with LDS counters (tf 2.0: 41 x 41 x 3 bins) and with global atomics
    (tf 13.36: 269 x 269 x 3) -- against the oracle, bit for bit, and the memory behind the bins stays untouched."""
    from icp_flow_amd import hist as hip_hist
    a = rp.default_args(max_points=300, translation_frame=tf, icp_max_iterations=2)
    ex, ey, ez = rp.bin_edges(a)
    lens = (len(ex), len(ey), len(ez))
    L = lens[0] * lens[1] * lens[2]
    S, D, _ = synthetic.make_batch(5, 300, seed=77)
    hi = np.nextafter(np.float32(ex.max()), np.float32(0.0))
    for b in (1, 2, 4):                       # v = dst_i - src_j = pred(max) on x (and on y for pair 2): p = len
        D[b, :4, :3] = (0.0, 0.0, 0.5)
        S[b, 0, :3] = (-hi, -(hi if b == 2 else ex[7].item()), 0.5)
    vx = np.float32(D[1, 0, 0]) - np.float32(S[1, 0, 0])
    assert vx < ex.max() and (vx - np.float32(ex.min())) / (np.float32(ex.max()) - np.float32(ex.min())) == np.float32(1.0)
    want = rp.hist(C(D), C(S), ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), *lens).numpy()
    clean_D = D.copy()
    clean_D[[1, 2, 4], :4, 3] = 0.0           # (flags off: only for the expectation below, through the all-flags vote)
    base = rp.hist(C(clean_D), C(S), ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), *lens).numpy()
    assert (want[2] - base[2]).reshape(-1)[:lens[1] * lens[2]].sum() >= 4      # pair 1's overflow sits in pair 2's first x row
    # public entry point
    got = hip_hist.hist(G(D), G(S), ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), *lens)
    assert np.array_equal(got.cpu().numpy(), want)
    # ... and through the C ABI into a buffer with canaries behind the [B, L] bins: the last pair's overflow is dropped
    out = torch.full((5 * L + 4096,), -7.0, dtype=torch.float32, device=DEV)
    d, s = G(D), G(S)
    _lib.call("icpflow_hist_vote", _lib.ptr(d), _lib.ptr(s), 5, 300, 300, float(ex.min()), float(ey.min()), float(ez.min()),
              float(ex.max()), float(ey.max()), float(ez.max()), lens[0], lens[1], lens[2], _lib.ptr(out), _lib.stream(DEV))
    assert np.array_equal(out[:5 * L].cpu().numpy().reshape(want.shape), want)
    assert bool((out[5 * L:] == -7.0).all())
    # fused vote (z-sorted kernel) through the debug export; the workspace region behind the bins is the library's own
    bins = torch.full((5, L), -1, dtype=torch.int32, device=DEV)
    with _lib.options(vote_bins=bins):
        utils_hist.estimate_init_pose(a, G(S), G(D))
    assert np.array_equal(bins.cpu().numpy().view(np.uint32).astype(np.int64), want.reshape(5, L).astype(np.int64))


# ------------------------------------------------------------------------------------------ ragged real shape
def test_ragged_real_shape_16_pairs_vs_oracle():
    """The shape real sweeps present (SURVEY 8(d) ragged variant, bench.py extras.ragged_real_shape): clusters of
    n ~ logUniform(20, 10^4) points padded to max_points = 10000, the reference's 100-iteration cap.  The first 16 pairs
    of the bench's batch against the oracle: initial poses equal, iteration count that of the exact evaluation, every
    pair against the exact (fp64-Kabsch) evaluation of the oracle, and against its fp32 evaluation wherever the two agree."""
    S, D, _ = synthetic.make_batch(128, 10000, seed=0, ragged=True, n_min=20)
    S, D = S[:16], D[:16]
    a = rp.default_args(max_points=10000, icp_max_iterations=100)
    _all_host_threads()
    T32, aux32 = rp.hist_icp(a, C(S), C(D), max_iterations=100, return_aux=True)
    T64, aux64 = rp.hist_icp(a, C(S), C(D), max_iterations=100, return_aux=True, kabsch_dtype=torch.float64, init=aux32["init"])
    T, iters = utils_match.hist_icp(a, G(S), G(D), return_iterations=True)
    init = utils_hist.estimate_init_pose(a, *[G(x) for x in _smaller_first(S, D)]).cpu().numpy()
    assert np.array_equal(init, aux32["init"].numpy())
    T = T.cpu().numpy()
    err32, err64 = displacement(T, T32.numpy(), S), displacement(T, T64.numpy(), S)
    agree = displacement(T32.numpy(), T64.numpy(), S) < DETERMINED_TOL
    n = [(int((S[b, :, 3] > 0).sum()), int((D[b, :, 3] > 0).sum())) for b in range(16)]
    msg = (f"cluster sizes {n}\niterations HIP {int(iters)}, fp32 oracle {aux32['iterations']}, fp64-Kabsch oracle {aux64['iterations']}\n"
           f"vs fp64-Kabsch oracle {np.round(err64, 7).tolist()}\nvs fp32 oracle {np.round(err32, 7).tolist()}\nfp32 and fp64 oracle agree on {agree.tolist()}")
    print(msg)
    assert int(iters) == aux64["iterations"], msg
    assert err64.max() < 1e-5, msg
    assert err32[agree].max() < TOL_M, msg


def _smaller_first(S, D):
    n1, n2 = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    sw = n1 > n2                                                                # utils_match.py:142
    A, B = S.copy(), D.copy()
    A[sw], B[sw] = D[sw], S[sw]
    return A, B


# ------------------------------------------------------------------------------------------ several batches in flight
def test_hist_icp_many_equals_separate_calls():
    """icpflow_hist_icp_many: K independent batches through one call, on internal worker streams forked from / joined
    into the caller's stream.  Every batch keeps its own batch-global stop: transforms and iteration counts are those of
    K separate hist_icp calls, bit for bit -- batches of different sizes, more batches than worker streams, called from
    the default and from a side stream, results consumed on the caller's stream without a host synchronisation."""
    N = 1024
    a = rp.default_args(max_points=N, icp_max_iterations=50)
    shapes = [(256, False, 0), (64, True, 300), (256, False, 256), (300, True, 600), (17, False, 900), (256, False, 512)]
    batches = [synthetic.make_batch(B, N, seed=0, first=first, ragged=r, n_min=40) for B, r, first in shapes]
    srcs, dsts = [G(b[0]) for b in batches], [G(b[1]) for b in batches]
    want = [utils_match.hist_icp(a, s_, d_, return_iterations=True) for s_, d_ in zip(srcs, dsts)]
    for stream in (None, torch.cuda.Stream(DEV)):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(DEV)):
            outs, iters = utils_match.hist_icp_many(a, srcs, dsts, return_iterations=True)
            total = torch.stack([o.sum() for o in outs]).sum()         # consumed on the same stream, no host sync
        torch.cuda.synchronize()
        for (T0, it0), T1, it1 in zip(want, outs, iters):
            assert int(it0) == int(it1) and torch.equal(T0, T1)
        assert torch.isfinite(total)
    # a single batch is the plain call
    one, = utils_match.hist_icp_many(a, srcs[:1], dsts[:1])
    assert torch.equal(one, want[0][0])


# ------------------------------------------------------------------------------------------ launch shape vs results
def test_results_do_not_depend_on_what_else_is_in_the_batch():
    """The ICP launch picks its workgroup shape from the batch (teams of workgroups for few large pairs, one 1024-thread
    workgroup per pair, two 512-thread workgroups per CU above two pairs per CU, ticket dispatch beyond the GPU's
    capacity), and the fp64 moment sums follow that shape in their last bits.  With the per-pair stop rule (no coupling
    through the batch-global stop) the same 40 pairs must come out the same whatever surrounds them: bounded by the
    rounding of one fp32 state, far below the north-star tolerance."""
    S, D, _ = synthetic.make_batch(1400, 2048, seed=3, ragged=True, n_min=600)
    a = rp.default_args(max_points=2048, icp_max_iterations=50, icp_stop_mode="per_pair")
    ref = None
    for B in (40, 300, 700, 1400):        # teams / one workgroup per pair / two per CU / ticket dispatch
        T = utils_match.hist_icp(a, G(S[:B]), G(D[:B]))[:40].cpu().numpy()
        assert np.isfinite(T).all()
        if ref is None:
            ref = T
        else:
            d = displacement(T, ref, S[:40])
            print(f"B {B}: max displacement of the first 40 pairs vs B 40: {d.max():.2e} m")
            assert d.max() < 2e-5
