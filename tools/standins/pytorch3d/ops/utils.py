import torch


def wmean(x, weight=None, dim=-2, keepdim=True, eps=1e-9):
    if weight is None:
        return x.mean(dim=dim, keepdim=keepdim)
    w = weight[..., None]
    return (x * w).sum(dim=dim, keepdim=keepdim) / w.sum(dim=dim, keepdim=keepdim).clamp(eps)


def eyes(dim, N, device=None, dtype=torch.float32):
    return torch.eye(dim, device=device, dtype=dtype)[None].repeat(N, 1, 1)


def is_pointclouds(x):
    return False


def convert_pointclouds_to_tensor(t):
    return t, torch.full((t.shape[0],), t.shape[1], dtype=torch.int64, device=t.device)
