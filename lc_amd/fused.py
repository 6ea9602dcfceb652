"""One launch for a batch of "pose units" (BASELINE metric): LC-loss forward+backward and weighted PnP on the same
B x N correspondences share a grid (`lc_pose_unit2_f32`: N <= 64 one wavefront per unit; dense shapes 256 < N <= 2048 tiled loss +
four-wave solve).  Pre-allocated outputs make the call allocation-free,
which is what bench.py times; results are bit-identical to the two stand-alone kernels (tests/test_gpu_fused.py)."""
from __future__ import annotations

import torch

from . import _lib


class PoseUnit:
    def __init__(self, B: int, N: int, device, want_pts3d: bool = True):
        lib = _lib.load()
        # dense shapes (256 < N <= 2048): the tiled loss and the four-wave solve share a grid (lc_pose_unit2_f32) where the loss has a
        # tiled form for (B, N); its workspace is zeroed once here and left zeroed by every launch
        self.ws = None
        if N > 64:
            nbytes = int(lib.lc_cov_loss_workspace_bytes(B, N)) if 256 < N <= 2048 else 0
            if nbytes == 0:
                raise ValueError("lc_amd.fused.PoseUnit needs N <= 64, or a dense shape whose loss takes the tiled form (256 < N <= 2048, "
                                 "B x ceil(N / 256) <= 256); use cov_mixed / pnp separately")
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        f = dict(device=device, dtype=torch.float32)
        self.B, self.N, self.device = B, N, device
        self.loss = torch.empty(B, **f)
        self.d_pts2d = torch.empty(B, N, 2, **f)
        self.d_inv_std = torch.empty(B, N, 2, **f)
        self.d_pts3d = torch.empty(B, N, 3, **f) if want_pts3d else None
        self.states = torch.empty(B, 7, **f)
        self.trust_radius = torch.empty(B, **f)
        self.invalid = torch.empty(B, device=device, dtype=torch.int32)
        self.iters = torch.empty(B, device=device, dtype=torch.int32)

    def __call__(self, K, pose, pts3d, pts2d, inv_std, bbox_3d, start, grad_out=None, valid=None, max_iter_count=50,
                 function_tolerance=1e-6, max_err_len=32, rel_thresh=3, w_e_thresh=4):
        """Inputs must already be contiguous float32 device tensors (checked once by the caller via prepare())."""
        lib = _lib.load()
        P = _lib.ptr
        rc = lib.lc_pose_unit2_f32(P(K), P(pose), P(pts3d), P(pts2d), P(inv_std), P(valid), P(bbox_3d), P(grad_out), self.B, self.N,
                                   float(max_err_len), float(rel_thresh), float(w_e_thresh), P(self.loss), P(self.d_pts2d),
                                   P(self.d_inv_std), P(self.d_pts3d), P(inv_std), P(start), P(self.states), P(self.trust_radius),
                                   P(self.invalid), P(self.iters), int(max_iter_count), float(function_tolerance), P(self.ws),
                                   0 if self.ws is None else self.ws.numel(), _lib.stream_ptr(self.device))
        _lib.check(rc, "lc_pose_unit2_f32")
        return self

    @staticmethod
    def prepare(**tensors):
        return {k: (None if v is None else _lib.require_hip_f32(k, v)) for k, v in tensors.items()}
