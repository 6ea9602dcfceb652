"""`lib.cov_mixed.Loss_cov_mixed` call surface on top of the fused HIP kernel.

Same signature, kwargs, output and gradient semantics as `lib/cov_mixed.py:100-150`; the ~1000 torch ops
of the reference become one `lc_cov_loss3_fwd_bwd_f32` launch (forward + unit Jacobians) plus, in backward,
one `lc_scale_rows_f32` launch applying the incoming cotangent.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


_TILED_WS = {}  # (device index, stream, B, N) -> zeroed workspace of the tiled form (lc_cov_loss3_fwd_bwd_f32)


def tiled_workspace(device, B: int, N: int):
    """The workspace that lets a dense sample (N > 256) spread over several compute units, or None when the shape does not use one.
    Eager launches: one per (device, stream, shape), zeroed once -- launches of one stream are ordered, so they can share it, and every
    launch leaves it zeroed.  Inside a hipGraph capture: a FRESH zeroed tensor per call, never cached -- it lives in the capturing
    graph's own pool (a cached one would tie later graphs to the first graph's memory), and its zero-fill is a node of the graph, so
    every replay starts from a clean workspace."""
    lib = _lib.load()
    nbytes = int(lib.lc_cov_loss_workspace_bytes(B, N))
    if nbytes == 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), _lib.raw_stream(device), B, N)
    ws = _TILED_WS.get(key)
    if ws is None:
        if len(_TILED_WS) >= 64:
            _TILED_WS.clear()  # a long-running process that walks through many shapes / streams: start over (entries are <= 4 MB each)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _TILED_WS[key] = ws
    return ws


def _launch_loss(K, pose, pts3d, pts2d, inv_std, valid, bbox, grad_out, max_err_len, rel_thresh, w_e_thresh,
                 want_grads: bool, want_pts3d: bool, want_aux: bool = False, cov_2d: bool = False, tiled: bool = True):
    """One fused launch.  Returns loss (B,), d_pts2d, d_inv_std, d_pts3d, aux (None where not requested)."""
    lib = _lib.load()
    B, N = pts3d.shape[0], pts3d.shape[1]
    loss = torch.empty(B, device=pts3d.device, dtype=torch.float32)
    d_u = torch.empty_like(pts2d) if want_grads else None
    d_s = torch.empty_like(inv_std) if want_grads else None
    d_x = torch.empty_like(pts3d) if (want_grads and want_pts3d) else None
    aux = torch.empty(B, 40, device=pts3d.device, dtype=torch.float32) if want_aux else None
    with _lib.on_device(pts3d.device):
        ws = tiled_workspace(pts3d.device, B, N) if (tiled and N > 256) else None  # tiled=False: the one-workgroup form (tests)
        rc = lib.lc_cov_loss3_fwd_bwd_f32(
            _lib.ptr(K), _lib.ptr(pose), _lib.ptr(pts3d), _lib.ptr(pts2d), _lib.ptr(inv_std), _lib.ptr(valid), _lib.ptr(bbox),
            _lib.ptr(grad_out), B, N, float(max_err_len), float(rel_thresh), float(w_e_thresh), int(cov_2d), _lib.ptr(loss), _lib.ptr(d_u),
            _lib.ptr(d_s), _lib.ptr(d_x), _lib.ptr(aux), _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.stream_ptr(pts3d.device))
    _lib.check(rc, "lc_cov_loss3_fwd_bwd_f32")
    return loss, d_u, d_s, d_x, aux


def _launch_scale(scale, srcs):
    lib = _lib.load()
    B = scale.shape[0]
    outs = [torch.empty_like(s) if s is not None else None for s in srcs]
    args = []
    for s, o in zip(srcs, outs):
        args += [_lib.ptr(s), _lib.ptr(o), 0 if s is None else s.numel() // B]
    with _lib.on_device(scale.device):
        rc = lib.lc_scale_rows_f32(_lib.ptr(scale), B, *args, _lib.stream_ptr(scale.device))
    _lib.check(rc, "lc_scale_rows_f32")
    return outs


class _LossCovMixedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, pose, pts3d, pts2d, inv_std, valid, bbox, max_err_len, rel_thresh, w_e_thresh, cov_2d=False):
        need = ctx.needs_input_grad
        want_grads = need[2] or need[3] or need[4]
        loss, d_u, d_s, d_x, _ = _launch_loss(K, pose, pts3d, pts2d, inv_std, valid, bbox, None, max_err_len, rel_thresh,
                                              w_e_thresh, want_grads, need[2], cov_2d=cov_2d)
        ctx.have = (d_x is not None, want_grads)
        saved = [t for t in (d_x, d_u, d_s) if t is not None]
        ctx.save_for_backward(*saved)
        return loss

    @staticmethod
    def backward(ctx, gout):
        have_x, have = ctx.have
        if not have:
            return (None,) * 11
        saved = list(ctx.saved_tensors)
        d_x = saved.pop(0) if have_x else None
        d_u, d_s = saved
        gout = gout.contiguous().to(torch.float32)
        g_x, g_u, g_s = _launch_scale(gout, [d_x, d_u, d_s])
        return None, None, g_x, g_u, g_s, None, None, None, None, None, None


def Loss_cov_mixed(K_out: Tensor, pose_gt: Tensor, pts3d: Tensor, pts2d_out: Tensor, inv_std2d: Tensor, valid_factor: Tensor,
                   **kwargs) -> Tensor:
    """Drop-in for `lib.cov_mixed.Loss_cov_mixed` (cov_mixed.py:100-150).

    kwargs: bbox_3d (required), max_err_len=32, rel_thresh=3, w_e_thresh=4, cov_2d=False (covariance of the projected
    bbox corners instead of the 3D ones, cov_mixed.py:125-130; no reference call site enables it -- losses.py:333,383).
    Gradients flow to pts2d_out, inv_std2d and pts3d; K_out / pose_gt / bbox_3d are treated as constants, as every
    reference caller passes ground-truth (non-differentiable) tensors there.
    """
    bbox_3d = kwargs["bbox_3d"]
    for name, t in (("K_out", K_out), ("pose_gt", pose_gt), ("bbox_3d", bbox_3d)):
        if t.requires_grad:
            raise NotImplementedError(f"lc_amd: gradient w.r.t. {name} is not provided by the fused kernel")
    if pts3d.dim() != 3:
        raise ValueError("lc_amd: Loss_cov_mixed expects batched inputs (B,N,*)")
    B, N = pts3d.shape[:2]
    args = dict(K_out=K_out.expand(B, 3, 3), pose_gt=pose_gt.expand(B, 7), pts3d=pts3d, pts2d_out=pts2d_out.expand(B, N, 2),
                inv_std2d=inv_std2d.expand(B, N, 2), bbox_3d=bbox_3d.expand(B, 8, 3))
    args = {k: _lib.require_hip_f32(k, v) for k, v in args.items()}
    valid = None
    if valid_factor is not None:
        valid = _lib.require_hip_f32("valid_factor", valid_factor.to(torch.float32).expand(B, N))
    return _LossCovMixedFn.apply(args["K_out"], args["pose_gt"], args["pts3d"], args["pts2d_out"], args["inv_std2d"], valid,
                                 args["bbox_3d"], kwargs.get("max_err_len", 32), kwargs.get("rel_thresh", 3),
                                 kwargs.get("w_e_thresh", 4), bool(kwargs.get("cov_2d", False)))


def loss_cov_mixed_fused(K, pose, pts3d, pts2d, inv_std, valid, bbox_3d, grad_out=None, want_pts3d=True, want_aux=False,
                         max_err_len=32, rel_thresh=3, w_e_thresh=4, cov_2d=False, tiled=True):
    """Non-autograd entry: loss AND input gradients for a known cotangent in ONE launch (bench / training loops that
    know d(total)/d(loss_b) up front, e.g. 1/B for `.mean()`).  Returns (loss, d_pts2d, d_inv_std, d_pts3d, aux)."""
    ts = [_lib.require_hip_f32(n, t) for n, t in (("K", K), ("pose", pose), ("pts3d", pts3d), ("pts2d", pts2d),
                                                   ("inv_std", inv_std), ("bbox_3d", bbox_3d))]
    valid = None if valid is None else _lib.require_hip_f32("valid", valid)
    grad_out = None if grad_out is None else _lib.require_hip_f32("grad_out", grad_out)
    return _launch_loss(ts[0], ts[1], ts[2], ts[3], ts[4], valid, ts[5], grad_out, max_err_len, rel_thresh, w_e_thresh,
                        True, want_pts3d, want_aux, cov_2d=cov_2d, tiled=tiled)
