"""Drop-ins for the reference's utils_icp.py."""
import torch

from . import _lib
from .utils_icp_pytorch3d import iterative_closest_point, stop_mode_of

ICP_MAX_ITERATIONS = 100        # utils_icp.py:54
ICP_RELATIVE_RMSE_THR = 1e-6    # utils_icp.py:55


def _icp_options(args):
    """The reference hard-codes max_iterations=100 / relative_rmse_thr=1e-6 (utils_icp.py:54-55);
    BASELINE's benchmark caps iterations at 50, so both are optional attributes of `args`."""
    return (int(getattr(args, "icp_max_iterations", ICP_MAX_ITERATIONS)),
            float(getattr(args, "icp_relative_rmse_thr", ICP_RELATIVE_RMSE_THR)),
            stop_mode_of(getattr(args, "icp_stop_mode", "reference")))


def pytorch3d_icp(args, src, dst):
    """utils_icp.py:50-73: ICP, then the column-vector 4x4 [[R^T, T],[0,0,0,1]]."""
    max_it, rel, stop = _icp_options(args)
    sol = iterative_closest_point(src, dst, init_transform=None, thres=args.thres_dist,
                                  max_iterations=max_it, relative_rmse_thr=rel,
                                  estimate_scale=False, allow_reflection=False, verbose=False,
                                  stop_mode=stop)
    Rs, ts = sol.RTs.R, sol.RTs.T
    Rts = torch.cat([Rs, ts[:, None, :]], dim=1)
    Rts = torch.cat([Rts.permute(0, 2, 1), Rts.new_zeros(len(ts), 1, 4)], dim=1)
    Rts[:, 3, 3] = 1.0
    return Rts


def apply_icp(args, src, dst, init_poses, return_iterations=False):
    """utils_icp.py:20-48 as one fused call: ICP from init_poses, compose, mean-NN-error
    check before/after, roll back where ICP did not help."""
    s = _lib.cloud(src, "src")
    d = _lib.cloud(dst, "dst")
    assert s.shape == d.shape
    B, N, _ = s.shape
    init = init_poses.to(device=s.device, dtype=torch.float32).contiguous()
    assert init.shape == (B, 4, 4)
    max_it, rel, stop = _icp_options(args)
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=s.device)
    iters = torch.empty((1,), dtype=torch.int32, device=s.device)   # always written by the call
    ws = _lib.workspace(s.device, _lib.workspace_bytes(B, N))
    _lib.call("icpflow_apply_icp", _lib.ptr(s), _lib.ptr(d), _lib.ptr(init), B, N, float(args.thres_dist),
              max_it, rel, stop, _lib.ptr(out), _lib.ptr(iters), _lib.ptr(ws), ws.numel(),
              _lib.stream(s.device), _lib.opt())
    return (out, iters) if return_iterations else out
