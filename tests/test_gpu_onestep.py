"""One-step (teacher-forced) conformance of the ICP iteration against the oracle, on EVERY pair.

A trajectory comparison cannot pin a pair that is still sliding when the batch rule stops: fifty iterations amplify
the rounding of the reference's own fp32 reductions.  One iteration cannot amplify anything.  So: take the
oracle's trace of the whole batch (`rp.iterative_closest_point(..., trace=True)`: the reference's fp32 arithmetic,
utils_icp_pytorch3d.py:153-213), and for a spread of iterations k hand the HIP path the ORACLE's state k
(`init_transform`, include/icpflow_hip.h d_icp_init_R / d_icp_init_T), let it run that one iteration, and compare what
it produces (first row of its `t_history`) with the oracle's state k + 1:

  * gate decisions: the number of gated correspondences (sum of the weights of :160-161) is EQUAL,
  * rotation within 1e-5, where the points end up within 1e-4 m (the north-star tolerance), rmse within 1e-5 m,

for ALL pairs -- no `determined` mask, no allowance by count.  Two kinds of (pair, step) are treated separately, both
ENUMERATED by the test from the oracle alone:

  * gate-critical: a query whose nearest neighbour sits within GATE_MARGIN of the gate radius, or whose two nearest
    targets are within GATE_MARGIN of each other while gated (either may change with the last bit of the moved point);
    there the count may differ by at most the number of such queries;
  * fp32-limited: steps on which the oracle's own fp32 Kabsch step (torch's summation order, fp32 SVD) is more than
    half the tolerance away from the SAME step evaluated in fp64 (`kabsch_dtype=torch.float64`: same state, same
    gated correspondences, same formulas) -- an ill-conditioned covariance, where the reference's rounding, not its
    algorithm, sets the answer.  There the fp32 comparison is not made.

EVERY step, enumerated or not, is also compared with that fp64 evaluation of the oracle's step, tightly (rotation
1e-6, moved points 1e-5 m): nothing is excused from it but the gate-critical steps whose counts really differ.  The
report says how many steps were checked and how many fell into each class.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from icp_flow_amd import synthetic  # noqa: E402
from icp_flow_amd.utils_icp_pytorch3d import SimilarityTransform, iterative_closest_point  # noqa: E402
from oracle import core as ocore  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

DEV = torch.device("cuda:0")
TOL_R = 1e-5
TOL_M = 1e-4
TOL_RMSE = 1e-5
TIGHT_R = 1e-6           # against the fp64 evaluation of the oracle's step
TIGHT_M = 1e-5
GATE_MARGIN = 1e-6       # metres: "a neighbour within 1 um of the gate"


def G(x):
    return (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(DEV)


def _host_threads():
    n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    ocore.set_num_threads(n)


def _oracle_trace(args, S, D, cap):
    """The roles, initial poses and ICP input exactly as hist_icp forms them (utils_match.py:139-150, utils_icp.py:21),
    then the oracle's ICP with its per-iteration trace."""
    _host_threads()
    src, dst = torch.from_numpy(S), torch.from_numpy(D)
    n1, n2 = (src[:, :, 3] > 0).sum(1), (dst[:, :, 3] > 0).sum(1)
    swap = n1 > n2
    a, b = src.clone(), dst.clone()
    a[swap], b[swap] = dst[swap], src[swap]
    init = rp.estimate_init_pose(args, a, b)
    moved = rp.transform_points_batch(a, init)
    sol = rp.iterative_closest_point(moved, b, thres=args.thres_dist, max_iterations=cap, trace=True)
    return moved, b, sol


def _steps(n):
    """First 5, every 5th, last 3 iterations of an n-iteration trace."""
    return sorted(set(list(range(min(5, n))) + list(range(5, n, 5)) + list(range(max(n - 3, 0), n))))


def _critical_queries(Xt, Y, valid, n_y, thres, chunk):
    """Per pair: the number of valid queries whose gate decision or gated neighbour hangs on the last bits of the moved
    point -- |d1 - thres| <= GATE_MARGIN, or d2 - d1 <= GATE_MARGIN with d1 <= thres + GATE_MARGIN (d1, d2: distances to
    the two nearest targets).  Evaluated in fp64 on the GPU from the fp32 points (test-side arithmetic only)."""
    B, N, _ = Xt.shape
    out = torch.zeros(B, dtype=torch.int64)
    for i0 in range(0, B, chunk):
        x = G(Xt[i0:i0 + chunk]).double()
        y = G(Y[i0:i0 + chunk, :, :3]).double()
        d = torch.cdist(x, y)                                           # [c, N, N] Euclidean, fp64
        live = (torch.arange(N, device=DEV)[None, :] < G(n_y[i0:i0 + chunk])[:, None])
        d = d.masked_fill(~live[:, None, :], float("inf"))
        two = torch.topk(d, 2, dim=2, largest=False).values
        d1, d2 = two[:, :, 0], two[:, :, 1]
        crit = ((d1 - thres).abs() <= GATE_MARGIN) | ((d2 - d1 <= GATE_MARGIN) & (d1 <= thres + GATE_MARGIN))
        out[i0:i0 + chunk] = (crit & G(valid[i0:i0 + chunk])).sum(1).cpu()
    return out


def _one_step_conformance(S, D, cap, chunk):
    args = rp.default_args(max_points=S.shape[1], icp_max_iterations=cap)
    moved, fixed, sol = _oracle_trace(args, S, D, cap)
    B, N, _ = moved.shape
    X0 = moved[:, :, :3]
    valid = moved[:, :, 3] > 0
    n_y = (fixed[:, :, 3] > 0).sum(1)
    gm, gf = G(moved), G(fixed)
    steps = _steps(sol.iterations)
    checked = enumerated = limited = 0
    worst = dict(R=0.0, m=0.0, rmse=0.0, R64=0.0, m64=0.0)
    ones = torch.ones(B)
    lines, failures = [], []
    p64 = X0.double()
    for k in steps:
        if k == 0:
            Rk, Tk = torch.eye(3)[None].repeat(B, 1, 1), torch.zeros(B, 3)
            Xt = X0                                                               # :140-147: Xt = X before the loop
        else:
            Rk, Tk = sol.history[k - 1][0], sol.history[k - 1][1]
            Xt = torch.bmm(X0, Rk) + Tk[:, None, :]                              # :177, :395 (the oracle's own Xt)
        R1, T1, rmse1, cnt1 = sol.history[k]
        got = iterative_closest_point(gm, gf, init_transform=SimilarityTransform(G(Rk), G(Tk), torch.ones(B, device=DEV)),
                                      thres=args.thres_dist, max_iterations=2)
        rec = got.t_history.records()[0].cpu()                                    # state after ONE iteration from state k
        Rg, Tg, rmseg, cntg = rec[:, 0:9].reshape(B, 3, 3), rec[:, 9:12], rec[:, 12], rec[:, 14].long()
        crit = _critical_queries(Xt, fixed, valid, n_y, args.thres_dist, chunk)
        # the same step of the oracle with its Kabsch step in fp64 (teacher-forced from the same state)
        o64 = rp.iterative_closest_point(moved, fixed, thres=args.thres_dist, max_iterations=1, trace=True,
                                         init_transform=(Rk, Tk, ones), kabsch_dtype=torch.float64)
        R64, T64, _, cnt64 = o64.history[0]
        assert torch.equal(cnt64, cnt1)                                       # same state, same gate decisions

        def moved_by(R, T):
            return torch.bmm(p64, R.double()) + T.double()[:, None, :]

        def far(Ra, Ta, Rb, Tb):
            return (Ra - Rb).abs().amax((1, 2)), ((moved_by(Ra, Ta) - moved_by(Rb, Tb)).abs().amax(2) * valid).amax(1)

        dc = (cntg - cnt1.long()).abs()
        dR, dm = far(Rg, Tg, R1, T1)              # HIP vs the oracle's fp32 step
        dR64, dm64 = far(Rg, Tg, R64, T64)        # HIP vs the oracle's step evaluated in fp64
        oR, om = far(R1, T1, R64, T64)            # the oracle's fp32 step vs its own fp64 evaluation
        dr = (rmseg - rmse1).abs()
        plain = crit == 0
        fp32_limited = plain & ((oR > TOL_R / 2) | (om > TOL_M / 2))
        strict = plain & ~fp32_limited
        same_count = dc == 0
        checked += B
        enumerated += int((~plain).sum())
        limited += int(fp32_limited.sum())
        for name, bad in (("gated count", plain & ~same_count), ("gated count beyond the enumerated queries", ~plain & (dc > crit)),
                          ("rotation vs fp32 oracle", strict & ~(dR <= TOL_R)), ("moved points vs fp32 oracle", strict & ~(dm <= TOL_M)),
                          ("rmse", strict & ~(dr <= TOL_RMSE)),
                          ("rotation vs fp64 evaluation", same_count & ~(dR64 <= TIGHT_R)),
                          ("moved points vs fp64 evaluation", same_count & ~(dm64 <= TIGHT_M))):
            for b in torch.nonzero(bad)[:, 0].tolist():
                failures.append(f"step {k} pair {b}: {name}: count {int(cntg[b])} vs {int(cnt1[b])} (enumerated {int(crit[b])}), "
                                f"vs fp32 oracle |dR| {float(dR[b]):.2e} moved {float(dm[b]):.2e} m |drmse| {float(dr[b]):.2e}; vs "
                                f"fp64 evaluation |dR| {float(dR64[b]):.2e} moved {float(dm64[b]):.2e} m; oracle fp32 vs fp64 "
                                f"|dR| {float(oR[b]):.2e} moved {float(om[b]):.2e} m")
        if strict.any():
            worst["R"] = max(worst["R"], float(dR[strict].max()))
            worst["m"] = max(worst["m"], float(dm[strict].max()))
            worst["rmse"] = max(worst["rmse"], float(dr[strict].max()))
        worst["R64"] = max(worst["R64"], float(dR64[same_count].max()))
        worst["m64"] = max(worst["m64"], float(dm64[same_count].max()))
        lines.append(f"step {k:2d}: gated counts equal on {int(same_count.sum())}/{B}, gate-critical pairs {int((~plain).sum())} "
                     f"(different count: {int((~plain & ~same_count).sum())}), fp32-limited {int(fp32_limited.sum())} "
                     f"(oracle fp32 vs fp64 up to {float(om.max()):.2e} m); vs fp32 oracle max |dR| {float(dR[strict].max()):.2e} "
                     f"moved {float(dm[strict].max()):.2e} m |drmse| {float(dr[strict].max()):.2e}; vs fp64 evaluation |dR| "
                     f"{float(dR64[same_count].max()):.2e} moved {float(dm64[same_count].max()):.2e} m")
    summary = (f"{checked} (pair, iteration) steps checked at iterations {steps} of the oracle's {sol.iterations}; {enumerated} "
               f"enumerated as gate-critical (margin {GATE_MARGIN:g} m), {limited} as fp32-limited (the oracle's fp32 step more than "
               f"half the tolerance from its fp64 evaluation); on the other {checked - enumerated - limited}: gated counts equal, "
               f"vs the fp32 oracle max |dR| {worst['R']:.2e}, moved points {worst['m']:.2e} m, |drmse| {worst['rmse']:.2e}; every "
               f"step with equal counts vs the fp64 evaluation: max |dR| {worst['R64']:.2e}, moved points {worst['m64']:.2e} m")
    print("\n".join(lines + [summary]))
    assert not failures, "\n".join(failures[:40] + [summary])
    # the enumerations stay a small minority (else a margin is wrong)
    assert enumerated * 20 <= checked and limited * 50 <= checked, summary


def test_config2_every_pair_one_step_from_the_oracle_state():
    """BASELINE config 2, the whole 256 x 1024 batch."""
    S, D, _ = synthetic.make_batch(256, 1024, seed=0)
    _one_step_conformance(S, D, 50, chunk=64)


def test_config4_sample_every_pair_one_step_from_the_oracle_state():
    """BASELINE config 4's shape: the first 64 pairs x 2048 points of rank 0's shard."""
    S, D, _ = synthetic.make_batch(1024, 2048, seed=0)
    _one_step_conformance(S[:64], D[:64], 50, chunk=16)


def test_ragged_team_batch_every_pair_one_step_from_the_oracle_state():
    """The shape real frames present: a dozen ragged pairs, 300 ... 4000 points padded to 4000 -- few enough
    for the ICP to run its large pairs as TEAMS of workgroups (several members per pair, partial moments exchanged every
    iteration, long probes), with both role orders in the batch."""
    S, D, _ = synthetic.make_batch(12, 4000, seed=11, ragged=True, n_min=300)
    S[::4], D[::4] = D[::4].copy(), S[::4].copy()
    ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    assert (ns > nd).any() and (ns < nd).any() and max(ns.max(), nd.max()) > 2500
    _one_step_conformance(S, D, 30, chunk=2)
