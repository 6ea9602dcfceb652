"""Batched pose-error metrics on the GPU: `lib/utils/evaluate.py:333-339 compute_pose_errors`
(= `lib/utils/error6d.py` add / adi / re / te) for a whole test set in one launch."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib
from .transforms import quaternion_rep_to_RT


def compute_pose_errors(R_est: Tensor, t_est: Tensor, R_gt: Tensor, t_gt: Tensor, pts: Tensor, pts_off: Tensor = None,
                        pts_cnt: Tensor = None, want_adi: bool = True) -> dict:
    """R_* (B,3,3), t_* (B,3) or (B,3,1), pts (M,3) [or packed (P,3) with per-pose pts_off/pts_cnt (B,) int32].
    Returns dict(adi, add, re, te) of (B,) tensors -- the keys of the reference's `compute_pose_errors`."""
    lib = _lib.load()
    Re = _lib.require_hip_f32("R_est", R_est)
    Rg = _lib.require_hip_f32("R_gt", R_gt)
    te = _lib.require_hip_f32("t_est", t_est.reshape(-1, 3))
    tg = _lib.require_hip_f32("t_gt", t_gt.reshape(-1, 3))
    P = _lib.require_hip_f32("pts", pts)
    B = Re.shape[0]
    dev = Re.device
    if pts_off is not None:
        pts_off = pts_off.to(device=dev, dtype=torch.int32).contiguous()
        pts_cnt = pts_cnt.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty(B, 4, device=dev, dtype=torch.float32)
    with _lib.on_device(dev):
        rc = lib.lc_pose_errors_f32(_lib.ptr(Re), _lib.ptr(te), _lib.ptr(Rg), _lib.ptr(tg), _lib.ptr(P), _lib.ptr(pts_off), _lib.ptr(pts_cnt),
                                    B, P.shape[0], int(want_adi), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "lc_pose_errors_f32")
    return dict(adi=out[:, 0], add=out[:, 1], re=out[:, 2], te=out[:, 3])


def pose_errors_from_states(states_est: Tensor, states_gt: Tensor, pts: Tensor, **kw) -> dict:
    """Same, from (B,7) quaternion representations (the output of `cer_solver.solve`, `test.py:174`)."""
    Re, te = quaternion_rep_to_RT(states_est)
    Rg, tg = quaternion_rep_to_RT(states_gt)
    return compute_pose_errors(Re, te, Rg, tg, pts, **kw)
