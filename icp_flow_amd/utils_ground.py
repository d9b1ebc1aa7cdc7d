"""Drop-in for the height-threshold part of the reference's ground removal (utils_ground.py:26-33).

`segment_ground` upstream ANDs this mask with Patchwork++ (third-party C++ vendored under
patchwork-plusplus/, CPU preprocessing, out of scope -- SURVEY 8(f) row 4); that half is not built."""
import numpy as np
import torch


def segment_ground_thres(args, points):
    """True = non-ground: z > range_z + ground_slack (utils_ground.py:27-30).  numpy or torch in, same out."""
    thr = args.range_z + args.ground_slack
    if isinstance(points, torch.Tensor):
        return ~(points[:, 2] <= thr)
    return ~(np.asarray(points)[:, 2] <= thr)


def segment_ground(args, points):
    raise NotImplementedError(
        "icp_flow_amd: segment_ground needs Patchwork++ (utils_ground.py:16-23, patchwork-plusplus/), which is "
        "out of scope; use segment_ground_thres or a precomputed non-ground mask")
