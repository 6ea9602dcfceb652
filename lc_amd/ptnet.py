"""Sparse keypoint head of `ptnet.py` on the fused HIP soft-argmax kernels.

  softargmax_2d_std(prob2d, clamp_std=False)   -- same function as `ptnet.py:100-115` (input: probabilities)
  softargmax_1d_cov(prob1d)                    -- `ptnet.py:85-97`
  spatial_softargmax_2d_std(logits)            -- `ptnet.py:59-66` fused: flatten -> softmax -> soft-argmax in one pass
  sparse_head(kpt_logits)                      -- the out_dict the sparse branch of `ptnet.forward` returns
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


def _fwd(x: Tensor, is_prob: bool):
    lib = _lib.load()
    H, W = x.shape[-2:]
    M = x.numel() // (H * W)
    mean = torch.empty(x.shape[:-2] + (2,), device=x.device, dtype=torch.float32)
    std = torch.empty_like(mean)
    stats = torch.empty(x.shape[:-2] + (4,), device=x.device, dtype=torch.float32)
    with _lib.on_device(x.device):
        rc = lib.lc_softargmax2d_fwd(_lib.ptr(x), _lib.MAP_DTYPES[x.dtype], M, H, W, int(is_prob), _lib.ptr(mean), _lib.ptr(std),
                                     _lib.ptr(stats), _lib.stream_ptr(x.device))
    _lib.check(rc, "lc_softargmax2d_fwd")
    return mean, std, stats


class _SoftArgmax2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, is_prob):
        mean, std, stats = _fwd(x, is_prob)
        ctx.is_prob = is_prob
        ctx.save_for_backward(x, mean, std, stats)
        ctx.mark_non_differentiable(stats)
        return mean, std, stats

    @staticmethod
    def backward(ctx, g_mean, g_std, _g_stats):
        x, mean, std, stats = ctx.saved_tensors
        lib = _lib.load()
        H, W = x.shape[-2:]
        M = x.numel() // (H * W)
        g_mean = torch.zeros_like(mean) if g_mean is None else g_mean.contiguous().to(torch.float32)
        g_std = torch.zeros_like(std) if g_std is None else g_std.contiguous().to(torch.float32)
        g_in = torch.empty_like(x)
        with _lib.on_device(x.device):
            rc = lib.lc_softargmax2d_bwd(_lib.ptr(x), _lib.MAP_DTYPES[x.dtype], _lib.ptr(mean), _lib.ptr(std), _lib.ptr(stats),
                                         _lib.ptr(g_mean), _lib.ptr(g_std), M, H, W, int(ctx.is_prob), _lib.ptr(g_in),
                                         _lib.stream_ptr(x.device))
        _lib.check(rc, "lc_softargmax2d_bwd")
        return g_in, None


def _clamp_std(std: Tensor) -> Tensor:
    small_std = std < 1
    return torch.where(small_std, torch.exp((std - 1) * small_std), std)  # ptnet.py:112-114


def softargmax_1d_cov(prob1d: Tensor):
    """`ptnet.softargmax_1d_cov` (ptnet.py:85-97): prob1d (*,N) -> mean (*,), cov (*,) -- the 2D kernel on a one-row map."""
    x = _lib.require_hip_map("prob1d", prob1d).unsqueeze(-2)
    mean, std, _ = _SoftArgmax2dFn.apply(x, True)
    return mean[..., 0], std[..., 0] ** 2 - 1e-6


def softargmax_2d_std(prob2d: Tensor, clamp_std: bool = False):
    """`ptnet.softargmax_2d_std` (ptnet.py:100-115): prob2d (*,H,W) -> mean (*,2) [x,y], std (*,2)."""
    x = _lib.require_hip_map("prob2d", prob2d)
    mean, std, _ = _SoftArgmax2dFn.apply(x, True)
    return mean, (_clamp_std(std) if clamp_std else std)


def spatial_softargmax_2d_std(logits: Tensor, clamp_std: bool = False):
    """Fused `kpt_logits.flatten(-2).softmax(-1).reshape_as(kpt_logits)` + `softargmax_2d_std` (ptnet.py:61)."""
    x = _lib.require_hip_map("kpt_logits", logits)
    mean, std, _ = _SoftArgmax2dFn.apply(x, False)
    return mean, (_clamp_std(std) if clamp_std else std)


def sparse_head(kpt_logits: Tensor) -> dict:
    """What `ptnet.forward` returns for the sparse branch (ptnet.py:59-66)."""
    pts2d, pts2d_std = spatial_softargmax_2d_std(kpt_logits)
    return {"pts2d": pts2d, "pts2d_std": pts2d_std}
