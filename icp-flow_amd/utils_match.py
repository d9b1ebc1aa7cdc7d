"""Drop-ins for the registration entry points of the reference's utils_match.py."""
import torch

from . import _lib
from .utils_hist import bin_edges
from .utils_icp import _icp_options


def hist_icp(args, src, dst, return_iterations=False):
    """utils_match.py:138-157 -- one full registration per cluster pair in ONE call:
    smaller-cloud-first swap, histogram initial pose, ICP with roll-back, inverse for the
    swapped pairs.  src, dst: float32 [B, max_points, 4] -> transforms float32 [B,4,4]."""
    s = _lib.cloud(src, "src")
    d = _lib.cloud(dst, "dst")
    assert s.shape == d.shape, "src and dst must share [B, max_points, 4]"
    B, N, _ = s.shape
    ex, ey, ez = bin_edges(args, s.device)
    lens = (len(ex), len(ey), len(ez))
    max_it, rel, stop = _icp_options(args)
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=s.device)
    iters = torch.zeros((1,), dtype=torch.int32, device=s.device)
    ws = _lib.workspace(s.device, _lib.workspace_bytes(B, N, lens))
    _lib.call("icpflow_hist_icp", _lib.ptr(s), _lib.ptr(d), B, N, _lib.ptr(ex), lens[0], _lib.ptr(ey),
              lens[1], _lib.ptr(ez), lens[2], float(args.thres_dist // 2), float(args.thres_dist), max_it,
              rel, stop, _lib.ptr(out), _lib.ptr(iters), _lib.ptr(ws), ws.numel(), _lib.stream(s.device))
    return (out, iters) if return_iterations else out


def match_eval(args, pcd1, pcd2, transformations):
    """utils_match.py:159-213 -> (errors, inliers, ratios, ious) [B,2], translations [B,3],
    rotations [B,3] (Euler ZYX degrees)."""
    a = _lib.cloud(pcd1, "pcd1")
    b = _lib.cloud(pcd2, "pcd2")
    assert a.shape == b.shape
    B, N, _ = a.shape
    T = transformations.to(device=a.device, dtype=torch.float32).contiguous()
    assert T.shape == (B, 4, 4)
    dev = a.device
    o2 = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(4)]
    o3 = [torch.empty((B, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    ws = _lib.workspace(dev, _lib.workspace_bytes(B, N))
    _lib.call("icpflow_match_eval", _lib.ptr(a), _lib.ptr(b), _lib.ptr(T), B, N, float(args.thres_dist),
              _lib.ptr(o2[0]), _lib.ptr(o2[1]), _lib.ptr(o2[2]), _lib.ptr(o2[3]), _lib.ptr(o3[0]),
              _lib.ptr(o3[1]), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
    return o2[0], o2[1], o2[2], o2[3], o3[0], o3[1]
