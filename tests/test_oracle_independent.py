"""oracle/oracle_core.c pinned INDEPENDENTLY of itself.

The golden vectors of the vote and of the K = 1 nearest neighbour were produced by the reference's Python calling
stand-ins that are backed by oracle_core.c (the CUDA extension and pytorch3d cannot run here), so comparing
oracle_core.c with those fixtures compares it with itself.  This file restates the two primitives a second time, in
vectorised numpy float32 straight from the reference's source lines, and compares bit for bit:

  * the vote: hist_cuda/cpp/hist_cuda_core.cuh:40-60 -- flag test, float32 differences, the [min, max) box test,
    floor((v - min) / (max - min) * float(len)) with IEEE float32 division, one count per vote;
  * K = 1 nearest neighbour as pytorch3d.ops.knn_points (call sites utils_helper.py:27, utils_icp_pytorch3d.py:154):
    squared distance dx*dx + dy*dy + dz*dz accumulated in that order with the products fused into the running sum
    (how nvcc contracts `dist += diff * diff`), first minimum on ties, pads at 1e8 scanned like any other point.

CPU only; inputs are the G1 / G3 fixtures' INPUT arrays (not their outputs) and seeded random clouds with random flags,
points exactly on bin edges and outside the box.
"""
import numpy as np
import torch

from conftest import load_golden
from oracle import core

F = np.float32


def np_hist_vote(X, Y, mins, maxs, lens):
    """hist_cuda_core.cuh:40-60 in numpy float32.  X, Y [B,N,4]; -> float32 [B,Lx,Ly,Lz].
    The reference does not clamp p: a quotient that rounds to 1.0 gives p = len, and the FLAT index b*L + p_x*Ly*Lz +
    p_y*Lz + p_z (:56) then addresses the next row -- for p_x = len_x the next PAIR's bins (one [B, L] allocation,
    hist_cuda.cu:59); past the last pair the reference writes out of bounds (undefined): dropped here."""
    X, Y = np.asarray(X, F), np.asarray(Y, F)
    mins, maxs = np.asarray(mins, F), np.asarray(maxs, F)
    lens = tuple(int(v) for v in lens)
    flen = np.asarray(lens, F)                               # __int2float_rd(len): exact below 2^24
    L = lens[0] * lens[1] * lens[2]
    out = np.zeros(len(X) * L, np.int64)
    for b in range(len(X)):
        x = X[b][X[b, :, 3] > F(0.0)][:, :3]                 # :42-44
        y = Y[b][Y[b, :, 3] > F(0.0)][:, :3]
        v = x[:, None, :] - y[None, :, :]                    # :46-48, float32
        assert v.dtype == F
        inside = ((v >= mins) & (v < maxs)).all(-1)          # :49, [) on every axis
        v = v[inside]
        q = (v - mins) / (maxs - mins)                       # :52-54, IEEE float32 division
        q = q * flen
        assert q.dtype == F
        p = np.floor(q).astype(np.int64)                     # __float2int_rd
        flat = b * L + (p[:, 0] * lens[1] + p[:, 1]) * lens[2] + p[:, 2]                           # :56
        flat = flat[flat < len(out)]
        out += np.bincount(flat, minlength=len(out))                                             # :57-58
    return out.astype(F).reshape((len(X),) + lens)


def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64; the sum is rounded once to float64
    and once to float32 (the double rounding can only matter on an exact float32 tie of a 53-bit sum)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def np_knn1(p1, p2, len1=None, len2=None):
    """pytorch3d knn_points(K=1) in numpy: -> (d2 float32 [B,N1], idx int64 [B,N1]); rows >= len1 get (0, 0)."""
    p1, p2 = np.asarray(p1, F)[:, :, :3], np.asarray(p2, F)[:, :, :3]
    B, N1, _ = p1.shape
    d2 = np.zeros((B, N1), F)
    idx = np.zeros((B, N1), np.int64)
    for b in range(B):
        l1 = N1 if len1 is None else int(len1[b])
        l2 = p2.shape[1] if len2 is None else int(len2[b])
        if l1 == 0 or l2 == 0:
            continue
        d = p1[b, :l1, None, :] - p2[b, None, :l2, :]        # float32 differences
        acc = d[..., 0] * d[..., 0]                          # float32 product
        acc = _fma32(d[..., 1], d[..., 1], acc)
        acc = _fma32(d[..., 2], d[..., 2], acc)
        idx[b, :l1] = np.argmin(acc, axis=1)                 # numpy: FIRST minimum
        d2[b, :l1] = acc[np.arange(l1), idx[b, :l1]]
    return d2, idx


def _random_clouds(seed, B, N, spread, flags="random"):
    r = np.random.default_rng(seed)
    X = np.zeros((B, N, 4), F)
    Y = np.zeros((B, N, 4), F)
    X[:, :, :3] = r.uniform(-spread, spread, (B, N, 3))
    Y[:, :, :3] = X[:, :, :3] + r.normal(0, 0.3, (B, N, 3)) + r.uniform(-1.5, 1.5, (B, 1, 3))
    if flags == "random":                                    # hist_cuda/test.py draws its flags at random too
        X[:, :, 3] = r.integers(0, 2, (B, N)); Y[:, :, 3] = r.integers(0, 2, (B, N))
    else:
        X[:, :, 3] = 1.0; Y[:, :, 3] = 1.0
    return X, Y


def test_vote_equals_independent_numpy_statement_on_fixture_inputs():
    g = load_golden("g1_hist_testpy")
    lens = [int(v) for v in g["lens"]]
    got = core.hist_vote(g["X"], g["Y"], g["mins"], g["maxs"], lens).numpy()
    want = np_hist_vote(g["X"], g["Y"], g["mins"], g["maxs"], lens)
    assert np.array_equal(got, want)
    assert [int(w.argmax()) for w in want] == [111987] * 3   # the reference's own known answer (hist_cuda/test.py)
    for tag in ("tf2p0", "tf3p34"):
        g = load_golden("g1_hist_ref_" + tag)
        lens = [int(v) for v in g["lens"]]
        got = core.hist_vote(g["dst"], g["src"], g["mins"], g["maxs"], lens).numpy()
        assert np.array_equal(got, np_hist_vote(g["dst"], g["src"], g["mins"], g["maxs"], lens))
        assert got.sum() > 0


def test_vote_equals_independent_numpy_statement_on_edges_and_random_flags():
    """Random flags, differences exactly on the box's min (votes), exactly on its max (does not), outside the box,
    on interior bin edges, duplicates; the reference-style box [edges[0], edges[L-1]) with L bins."""
    ex = torch.arange(-2.0, 2.0 + 0.1 - 1e-8, 0.1).numpy()   # utils_hist.py:63-65 (float32 arange)
    ez = torch.arange(-0.1, 0.1 + 0.1 - 1e-8, 0.1).numpy()
    mins, maxs, lens = (ex.min(), ex.min(), ez.min()), (ex.max(), ex.max(), ez.max()), (len(ex), len(ex), len(ez))
    X, Y = _random_clouds(7, 5, 300, 3.0)
    # hand-made rows of pair 0: X_i - Y_j lands exactly on edges
    X[0, :8, :3] = 0.0
    X[0, :8, 3] = 1.0
    Y[0, 0, :3] = (-ex[0], -ex[0], -ez[0])                   # v = (min, min, min): votes, bin (0, 0, 0)
    Y[0, 1, :3] = (-ex[-1], 0.0, 0.0)                        # v_x = max: outside
    Y[0, 2, :3] = (-ex[7], -ex[19], -ez[1])                  # interior edges
    Y[0, 3, :3] = (-ex[7], -ex[19], -ez[1])                  # duplicate
    Y[0, 4, :3] = (5.0, 0.0, 0.0)                            # outside
    Y[0, 5, :3] = (np.nextafter(F(-ex[-1]), F(0.0)), 0.0, 0.0)   # just inside max
    Y[0, :6, 3] = 1.0
    # v_x = pred(max): (v - min) rounds to (max - min), the quotient to 1.0, p_x = len_x -> the NEXT pair's bins
    # (pairs 2 -> 3); for the last pair (4) the vote is dropped
    for b in (2, 4):
        X[b, :3, :3] = 0.0
        X[b, :3, 3] = 1.0
        Y[b, 0, :3] = (-np.nextafter(F(ex[-1]), F(0.0)), -ex[11], -ez[1])
        Y[b, 0, 3] = 1.0
    got = core.hist_vote(X, Y, mins, maxs, lens).numpy()
    want = np_hist_vote(X, Y, mins, maxs, lens)
    assert np.array_equal(got, want)
    assert want[0, 0, 0, 0] >= 8 and want.sum() > 1000
    vx = F(0.0) - Y[2, 0, 0]
    assert vx < maxs[0] and (vx - F(mins[0])) / (F(maxs[0]) - F(mins[0])) == F(1.0)       # the case really overflows
    X2, Y2 = X.copy(), Y.copy()
    X2[2, :3, 3] = 0.0                                          # without those rows pair 3's bin (0, 11, 1) holds 3 fewer
    assert (want[3] - np_hist_vote(X2, Y2, mins, maxs, lens)[3])[0, 11, 1] == 3.0
    # another box: non-symmetric, lengths that do not divide the range
    mins2, maxs2, lens2 = (F(-1.37), F(-0.2), F(-0.33)), (F(2.11), F(3.7), F(0.41)), (23, 57, 5)
    assert np.array_equal(core.hist_vote(X, Y, mins2, maxs2, lens2).numpy(), np_hist_vote(X, Y, mins2, maxs2, lens2))


def test_knn_equals_independent_numpy_statement():
    g = load_golden("g3_nn")                                  # padded pairs, pads at 1e8, n_s != n_d
    for a, b in ((g["src"], g["dst"]), (g["dst"], g["src"])):
        d2, idx, _ = core.knn1(a[:, :, :3], b[:, :, :3])      # un-lengthed, pads included (utils_helper.py:27)
        wd, wi = np_knn1(a, b)
        assert np.array_equal(idx.numpy(), wi) and np.array_equal(d2.numpy(), wd)
        # with valid prefixes (utils_icp_pytorch3d.py:154-156)
        la, lb = (a[:, :, 3] > 0).sum(1), (b[:, :, 3] > 0).sum(1)
        d2, idx, _ = core.knn1(a[:, :, :3], b[:, :, :3], la, lb)
        wd, wi = np_knn1(a, b, la, lb)
        assert np.array_equal(idx.numpy(), wi) and np.array_equal(d2.numpy(), wd)
    # exact ties: lattice points and duplicated targets -> the FIRST minimum
    r = np.random.default_rng(3)
    P = np.zeros((3, 200, 4), F)
    Q = np.zeros((3, 240, 4), F)
    P[:, :, :3] = np.floor(r.uniform(-2, 2, (3, 200, 3)) * 8) / 8
    Q[:, :, :3] = np.floor(r.uniform(-2, 2, (3, 240, 3)) * 8) / 8
    Q[:, 100:140, :3] = Q[:, 0:40, :3]                        # duplicates later in the array must lose
    Q[1, 200:] = (1e8, 1e8, 1e8, 0.0)
    d2, idx, _ = core.knn1(P[:, :, :3], Q[:, :, :3])
    wd, wi = np_knn1(P, Q)
    assert np.array_equal(idx.numpy(), wi) and np.array_equal(d2.numpy(), wd)
    assert (np.diff(np.sort(wd, axis=1), axis=1) == 0).any()  # the case really holds ties
