"""Drop-in for the reference's utils_flow.flow_estimation_torch (utils_flow.py:57-69)."""
import torch

from . import _lib


def flow_estimation_torch(args, src_points, dst_points, src_labels, dst_labels, pairs, transformations, pose):
    """Per-point flow: points of a matched cluster move with T_cluster @ pose, all others with
    pose alone; flow = moved - point.  The reference builds an N x P label-equality matrix and two
    N-batched 4x4 bmm; here one thread per point looks its label up and applies one 3x4 map."""
    assert len(src_points) == len(src_labels)
    pts = src_points[:, 0:3].contiguous().float()
    _lib.require_gpu(pts, src_labels)
    dev = pts.device
    N = pts.shape[0]
    P = int(pairs.shape[0])
    lab = src_labels.contiguous().float()
    flow = torch.empty((N, 3), dtype=torch.float32, device=dev)
    if P:
        # the matched source labels are read in place: column 0 of the float32 pair rows (no slice kernel)
        rows = pairs if (pairs.device == dev and pairs.dtype == torch.float32 and pairs.dim() == 2 and pairs.stride(1) == 1
                         and 1 <= pairs.stride(0) <= 1024) else pairs[:, 0:1].to(dev).contiguous().float()
        stride = int(rows.stride(0)) if rows.shape[1] > 1 else 1
        T = transformations.to(dev).contiguous().float()
    else:
        rows, stride, T = None, 1, None
    pose = pose.to(dev).contiguous().float()
    _lib.call("icpflow_flow_rigid_rows", _lib.ptr(pts), _lib.ptr(lab), N, _lib.ptr(rows), stride, _lib.ptr(T), P,
              _lib.ptr(pose), _lib.ptr(flow), _lib.stream(dev))
    return flow
