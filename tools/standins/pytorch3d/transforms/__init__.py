import torch


def matrix_to_euler_angles(M, convention):
    assert convention == "ZYX"
    return torch.stack([torch.atan2(M[..., 1, 0], M[..., 0, 0]),
                        torch.asin(-M[..., 2, 0]),
                        torch.atan2(M[..., 2, 1], M[..., 2, 2])], dim=-1)
