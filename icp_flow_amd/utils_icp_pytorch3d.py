"""Drop-in for the reference's utils_icp_pytorch3d.py (its modified pytorch3d ICP)."""
from typing import List, NamedTuple, Union

import torch

from . import _lib


class SimilarityTransform(NamedTuple):          # utils_icp_pytorch3d.py:23-26
    R: torch.Tensor
    T: torch.Tensor
    s: torch.Tensor


class ICPSolution(NamedTuple):                  # utils_icp_pytorch3d.py:29-34
    converged: bool
    rmse: Union[torch.Tensor, None]
    Xt: torch.Tensor
    RTs: SimilarityTransform
    t_history: List[SimilarityTransform]


_STOP = {"reference": _lib.STOP_REFERENCE, "per_pair": _lib.STOP_PER_PAIR,
         _lib.STOP_REFERENCE: _lib.STOP_REFERENCE, _lib.STOP_PER_PAIR: _lib.STOP_PER_PAIR}


def stop_mode_of(mode):
    try:
        return _STOP[mode]
    except KeyError:
        raise ValueError(f"unknown ICP stop mode {mode!r} (use 'reference' or 'per_pair')") from None


def iterative_closest_point(X, Y, init_transform=None, thres=0.1, max_iterations=100,
                            relative_rmse_thr=1e-6, estimate_scale=False, allow_reflection=False,
                            verbose=False, stop_mode="reference"):
    """utils_icp_pytorch3d.py:37-225 on [B,N,4] clouds (x,y,z,flag).

    `init_transform` (SimilarityTransform with unit scale, :118-138): the first correspondence search runs on
    X R0 + T0.  `t_history` (:187) is materialised lazily from the per-iteration records on the device (reference
    stop rule, max_iterations <= 128, at most 64 MiB of records; empty otherwise) and `converged` is a lazy device
    flag.  `allow_reflection=True` (:354-362) returns the best orthogonal matrix instead of the best rotation.
    `estimate_scale=True` (:364-374): s = trace(E S) / Xcov -- trace(E S) is the largest eigenvalue of the closed-form
    solve -- and Xt = s X R + T.  Limit: similarity transforms (estimate_scale, or an init_transform whose scale is not
    1) are built for the sorted-sweep correspondence search only, i.e. N <= 16384 (below 64 points a request for a scale
    selects the sweep instead of the all-pairs scan, round 4) without a `search` override and
    with the default fp64 arithmetic; outside of that icpflow_icp refuses the call (RuntimeError, nothing enqueued).
    ICP-Flow itself never asks for either (utils_icp.py:51-58).
    """
    x = _lib.cloud(X, "X")
    y = _lib.cloud(Y, "Y")
    if x.shape != y.shape:
        raise ValueError("Point sets X and Y have to have the same number of batches, points and dimensions.")
    B, N, _ = x.shape
    dev = x.device
    init = None
    if init_transform is not None:
        try:                                                                        # :121-133
            R0, T0, s0 = init_transform
            assert R0.shape == (B, 3, 3) and T0.shape == (B, 3) and s0.shape == (B,)
        except Exception:
            raise ValueError("The initial transformation init_transform has to be a named tuple SimilarityTransform "
                             "with elements (R, T, s). R are dim x dim orthonormal matrices of shape (minibatch, dim, "
                             "dim), T is a batch of dim-dimensional translations of shape (minibatch, dim) and s is a "
                             "batch of scalars of shape (minibatch,).") from None
        init = (R0.to(device=dev, dtype=torch.float32).contiguous(), T0.to(device=dev, dtype=torch.float32).contiguous())
        if not bool((s0 == 1).all()):        # a scale other than 1 travels along (the similarity kernels)
            init = init + (s0.to(device=dev, dtype=torch.float32).contiguous(),)
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    T = torch.empty((B, 3), dtype=torch.float32, device=dev)
    rmse = torch.empty((B,), dtype=torch.float32, device=dev)
    flags = torch.zeros((2,), dtype=torch.int32, device=dev)        # [iterations, converged]
    hist = None
    if stop_mode_of(stop_mode) == _lib.STOP_REFERENCE and 2 <= int(max_iterations) <= 128 and \
            int(max_iterations) * B * 64 <= (64 << 20) and not (_lib._current()[-1]["flags"] & _lib.OPT_FLAGS["no_speculative"]) \
            and _lib._current()[-1]["arith"] == 0:
        hist = torch.empty((int(max_iterations), B, 16), dtype=torch.float32, device=dev)
    ws = _lib.workspace(dev, _lib.workspace_bytes(B, N))
    # the scale: estimated (:364-374), else 1 -- the scale of an initial transform shapes the first search only (:376-379)
    scale = torch.ones(B, dtype=torch.float32, device=dev)
    with _lib.options(icp_init=init, icp_history=hist, icp_allow_reflection=bool(allow_reflection),
                      icp_scale=scale if estimate_scale else None):
        _lib.call("icpflow_icp", _lib.ptr(x), _lib.ptr(y), None, B, N, float(thres), int(max_iterations),
                  float(relative_rmse_thr), stop_mode_of(stop_mode), _lib.ptr(R), _lib.ptr(T), _lib.ptr(rmse),
                  _lib.ptr(flags[0:1]), _lib.ptr(flags[1:2]), _lib.ptr(ws), ws.numel(), _lib.stream(dev), _lib.opt())
    # Xt = s X R + T (utils_icp_pytorch3d.py:177, :395) -- returned for API parity
    Xt = scale[:, None, None] * torch.bmm(x[:, :, 0:3], R) + T[:, None, :]
    sol = ICPSolution(_LazyFlag(flags, 1), rmse, Xt, SimilarityTransform(R, T, scale), _LazyHistory(hist, flags))
    return sol


class _LazyHistory(list):
    """t_history: one SimilarityTransform per executed iteration, built from the device records on first use."""

    def __init__(self, hist, flags):
        super().__init__()
        self._hist, self._flags, self._done = hist, flags, hist is None

    def _fill(self):
        if not self._done:
            self._done = True
            n = int(self._flags[0].item())
            h = self._hist[:max(n, 0)]
            B = h.shape[1]
            for k in range(h.shape[0]):
                super().append(SimilarityTransform(h[k, :, 0:9].reshape(B, 3, 3), h[k, :, 9:12], h[k, :, 13]))
        return self

    def records(self):
        """The raw per-iteration records [iterations, B, 16] on the device: R (9, row-major), T (3), rmse, scale, the
        number of gated correspondences (sum of the weights of :161) and one unused float; None when no history was kept."""
        if self._hist is None:
            return None
        return self._hist[:max(int(self._flags[0].item()), 0)]

    def __len__(self):
        return list.__len__(self._fill())

    def __iter__(self):
        return list.__iter__(self._fill())

    def __getitem__(self, i):
        return list.__getitem__(self._fill(), i)


class _LazyFlag:
    """bool() reads one int from the device (the only sync, and only if somebody asks)."""

    def __init__(self, flags, k):
        self._flags, self._k = flags, k

    def __bool__(self):
        return bool(int(self._flags[self._k].item()))

    @property
    def iterations(self):
        return int(self._flags[0].item())

    def __repr__(self):
        return f"{bool(self)}"
