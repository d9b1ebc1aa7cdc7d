"""Drop-in for the reference's `hist_cuda.hist` module (hist_cuda/hist.py:39-51).

    from icp_flow_amd.hist import hist      # instead of: from hist_cuda.hist import hist

The reference JIT-builds a torch C++/CUDA extension at import time (hist.py:14-37) and
exposes `HIST.hist(X, Y, min_x, min_y, min_z, max_x, max_y, max_z, len_x, len_y, len_z,
mini_batch) -> Tensor`.  Here the same call goes through the C ABI of libicpflow_hip.so.
"""
import torch

from . import _lib


def hist(X, Y, min_x, min_y, min_z, max_x, max_y, max_z, len_x, len_y, len_z, mini_batch=8):
    """Translation histogram of all valid (X_i - Y_j) -> float32 [B, len_x, len_y, len_z].

    Same contract as the reference extension (hist_cuda.cu:29-43): contiguous GPU tensors,
    equal batch, last dim 4 (x, y, z, flag).  The min/max scalars may be Python floats or
    0-dim tensors (the reference passes `bins.min()` device tensors, utils_hist.py:70-71,
    which pybind narrows to C float): they are narrowed to float32 the same way.
    `mini_batch` only chunked the reference's launches and is ignored.
    """
    _lib.require_gpu(X, Y)
    if not (X.is_contiguous() and Y.is_contiguous()):
        raise RuntimeError("input tensor has to be contiguous")          # hist_cuda.cu:29-30
    if X.dim() != 3 or Y.dim() != 3 or X.shape[2] != 4 or Y.shape[2] != 4:
        raise RuntimeError("dim != 4; 3 for (x, y, z); 1 for indicator, padded or not.")   # :43
    if X.shape[0] != Y.shape[0]:
        raise RuntimeError(f"batch_X ({X.shape[0]}) != batch_Y ({Y.shape[0]}).")          # :40
    if X.dtype != torch.float32 or Y.dtype != torch.float32:
        raise RuntimeError("icp_flow_amd.hist: float32 only (the reference only ever passes float)")
    B, NX, NY = X.shape[0], X.shape[1], Y.shape[1]
    lx, ly, lz = int(len_x), int(len_y), int(len_z)
    out = torch.empty((B, lx, ly, lz), dtype=torch.float32, device=X.device)
    _lib.call("icpflow_hist_vote", _lib.ptr(X), _lib.ptr(Y), B, NX, NY,
              float(min_x), float(min_y), float(min_z), float(max_x), float(max_y), float(max_z),
              lx, ly, lz, _lib.ptr(out), _lib.stream(X.device))
    return out
