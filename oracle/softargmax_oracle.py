"""CPU oracle for the sparse keypoint head (TEST INFRASTRUCTURE ONLY): plain-torch restatement of
`ptnet.py:59-66` (spatial softmax) and `ptnet.py:85-115` (softargmax_1d_cov / softargmax_2d_std).
Pinned by tests/test_oracle_head.py against tests/golden/head_*.npz (generated from the reference)."""
import torch
from torch import Tensor


def softargmax_1d_cov(prob1d: Tensor):
    """ptnet.py:85-97: mean = sum_i i p_i, cov = sum_i (i-mean)^2 p_i."""
    xx = torch.arange(prob1d.shape[-1], dtype=prob1d.dtype, device=prob1d.device)
    mean = (prob1d * xx).sum(-1)
    cov = (prob1d * (xx - mean.unsqueeze(-1)) ** 2).sum(-1)
    return mean, cov


def softargmax_2d_std(prob2d: Tensor):
    """ptnet.py:100-115 (clamp_std=False)."""
    mx, cx = softargmax_1d_cov(prob2d.sum(dim=-2))
    my, cy = softargmax_1d_cov(prob2d.sum(dim=-1))
    return torch.stack((mx, my), -1), (torch.stack((cx, cy), -1) + 1e-6).sqrt()


def spatial_softargmax_2d_std(logits: Tensor):
    """ptnet.py:61."""
    return softargmax_2d_std(logits.flatten(start_dim=-2).softmax(dim=-1).reshape_as(logits))
