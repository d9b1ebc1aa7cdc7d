"""Drop-ins for the reference's utils_hist.py (translation histogram -> initial pose)."""
import torch

from . import _lib
from .hist import hist  # noqa: F401  (re-exported like the reference module does)

_edge_cache = {}


def bin_edges(args, device=None):
    """Left bin edges exactly as utils_hist.py:61-65 builds them: float32 torch.arange on
    the host.  Returns (ex, ey, ez) CPU tensors, or device copies (cached) if `device`."""
    tf, th, eps = float(args.translation_frame), float(args.thres_dist), 1e-8
    key = (tf, th, None if device is None else (device.type, device.index))
    hit = _edge_cache.get(key)
    if hit is None:
        ex = torch.arange(-tf, tf + th - eps, th, dtype=torch.float32, device="cpu")
        ey = torch.arange(-tf, tf + th - eps, th, dtype=torch.float32, device="cpu")
        ez = torch.arange(-th, th + th - eps, th, dtype=torch.float32, device="cpu")
        hit = (ex, ey, ez) if device is None else tuple(e.to(device) for e in (ex, ey, ez))
        if len(_edge_cache) > 64:
            _edge_cache.clear()
        _edge_cache[key] = hit
    return hit


def topk_nms(x, k=5, kernel_size=11):
    """utils_hist.py:21-29 -> (votes [B,k], flat idx int64 [B,k]).  Equal votes are ordered
    by ascending flat index (torch.topk leaves that order unspecified)."""
    _lib.require_gpu(x)
    if x.dim() != 4 or x.dtype != torch.float32:
        raise RuntimeError("topk_nms: expected a float32 [B,H,W,D] histogram")
    x = x.contiguous()
    b, h, w, d = x.shape
    votes = torch.empty((b, k), dtype=torch.float32, device=x.device)
    idxs = torch.empty((b, k), dtype=torch.int64, device=x.device)
    need = 2 * (b * h * w * d * 4 + 256)
    ws = _lib.workspace(x.device, need)
    _lib.call("icpflow_hist_peaks", _lib.ptr(x), b, h, w, d, int(k), int(kernel_size), _lib.ptr(votes),
              _lib.ptr(idxs), _lib.ptr(ws), ws.numel(), _lib.stream(x.device))
    return votes, idxs


def estimate_init_pose_batch(args, src, dst):
    """utils_hist.py:46-124, one fused call (vote, NMS + top-5, 6-candidate scoring)."""
    s = _lib.cloud(src, "src")
    d = _lib.cloud(dst, "dst")
    assert s.shape == d.shape, "src and dst must share [B, max_points, 4]"
    B, N, _ = s.shape
    ex, ey, ez = bin_edges(args, s.device)
    lens = (len(ex), len(ey), len(ez))
    T = torch.empty((B, 4, 4), dtype=torch.float32, device=s.device)
    _lib.check_vote_bins(B, lens)
    ws = _lib.workspace(s.device, _lib.workspace_bytes(B, N, lens))
    shift = float(args.thres_dist // 2)                                  # utils_hist.py:78
    _lib.call("icpflow_estimate_init_pose", _lib.ptr(s), _lib.ptr(d), B, N, _lib.ptr(ex), lens[0],
              _lib.ptr(ey), lens[1], _lib.ptr(ez), lens[2], shift, _lib.ptr(T), _lib.ptr(ws),
              ws.numel(), _lib.stream(s.device), _lib.opt())
    return T


def estimate_init_pose(args, src, dst):
    """utils_hist.py:33-44.  The reference chunks by args.chunk_size only to bound memory
    (comment :31-32); results do not depend on it, so the whole batch goes in one call."""
    assert len(src) == len(dst)
    return estimate_init_pose_batch(args, src, dst)
