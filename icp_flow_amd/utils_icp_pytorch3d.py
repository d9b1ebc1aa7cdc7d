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

    Differences from the reference object, all outside what its callers read
    (utils_icp.py:60-61 reads RTs.R / RTs.T only): `t_history` is empty (the per-iteration
    transforms never leave the device) and `converged` is materialised lazily from a device
    flag.  estimate_scale / allow_reflection / init_transform are fixed to the values the
    reference passes (utils_icp.py:51-58) and anything else raises.
    """
    if estimate_scale or allow_reflection or init_transform is not None:
        raise NotImplementedError("only the configuration used by ICP-Flow (utils_icp.py:51-58) is built")
    x = _lib.cloud(X, "X")
    y = _lib.cloud(Y, "Y")
    if x.shape != y.shape:
        raise ValueError("Point sets X and Y have to have the same number of batches, points and dimensions.")
    B, N, _ = x.shape
    dev = x.device
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    T = torch.empty((B, 3), dtype=torch.float32, device=dev)
    rmse = torch.empty((B,), dtype=torch.float32, device=dev)
    flags = torch.zeros((2,), dtype=torch.int32, device=dev)        # [iterations, converged]
    ws = _lib.workspace(dev, _lib.workspace_bytes(B, N))
    _lib.call("icpflow_icp", _lib.ptr(x), _lib.ptr(y), None, B, N, float(thres), int(max_iterations),
              float(relative_rmse_thr), stop_mode_of(stop_mode), _lib.ptr(R), _lib.ptr(T), _lib.ptr(rmse),
              _lib.ptr(flags[0:1]), _lib.ptr(flags[1:2]), _lib.ptr(ws), ws.numel(), _lib.stream(dev), _lib.opt())
    # Xt = s X R + T (utils_icp_pytorch3d.py:177, :395) -- returned for API parity
    Xt = torch.baddbmm(T[:, None, :], x[:, :, 0:3], R)
    sol = ICPSolution(_LazyFlag(flags, 1), rmse, Xt,
                      SimilarityTransform(R, T, torch.ones(B, dtype=torch.float32, device=dev)), [])
    return sol


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
