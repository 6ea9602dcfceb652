"""Seeded synthetic 2D-3D correspondence batches of the reference's shape.

Follows SURVEY.md section 8(d): poses in mm (BOP units), keypoint extents of LM-O object 1
(`assets/fps/lmo.pkl`), crop-scaled intrinsics with in-plane rotation (`dataset.py:402-423`),
bbox corner order of `model_transform.py:6-18`.  Everything is generated on the CPU from one
`torch.Generator` so CPU oracle and HIP path see identical bits.
"""
from __future__ import annotations

import math
import torch

EXTENT_MM = (37.8, 37.9, 45.8)


def _quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def _quat_to_R(q):
    q = q / q.norm(dim=-1, keepdim=True)
    r, i, j, k = q.unbind(-1)
    o = torch.stack((1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r),
                     2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r),
                     2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def bbox3d_from_scale(scale: torch.Tensor) -> torch.Tensor:
    """8 corners in the order of `model_transform.py:6-18`."""
    signs = torch.tensor([[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1],
                          [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]], dtype=scale.dtype)
    return signs * scale


def make_batch(B: int, N: int, seed: int = 0, dtype=torch.float32, outlier_frac: float = 0.05,
               rotate_K: bool = True, noise_px: float = 1.0):
    """Returns a dict of CPU tensors: K, pose, pts3d, pts2d, inv_std, bbox_3d, start (perturbed pose for PnP)."""
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    q = torch.randn(B, 4, generator=g, dtype=f64)
    q = q / q.norm(dim=-1, keepdim=True)
    q = torch.where(q[:, :1] < 0, -q, q)
    t = torch.stack((torch.rand(B, generator=g, dtype=f64) * 100 - 50,
                     torch.rand(B, generator=g, dtype=f64) * 100 - 50,
                     torch.rand(B, generator=g, dtype=f64) * 600 + 600), -1)
    ext = torch.tensor(EXTENT_MM, dtype=f64)
    X = (torch.rand(B, N, 3, generator=g, dtype=f64) * 2 - 1) * ext
    f = torch.rand(B, generator=g, dtype=f64) * 100 + 200
    th = torch.rand(B, generator=g, dtype=f64) * (2 * math.pi) if rotate_K else torch.zeros(B, dtype=f64)
    K = torch.zeros(B, 3, 3, dtype=f64)
    K[:, 0, 0] = f * th.cos()
    K[:, 0, 1] = -f * th.sin()
    K[:, 1, 0] = f * th.sin()
    K[:, 1, 1] = f * th.cos()
    K[:, 0, 2] = 32
    K[:, 1, 2] = 32
    K[:, 2, 2] = 1
    R = _quat_to_R(q)
    Xc = X @ R.mT + t[:, None]
    xf = Xc @ K.mT
    proj = xf[..., :2] / xf[..., 2:3]
    noise = torch.randn(B, N, 2, generator=g, dtype=f64) * noise_px
    outl = torch.rand(B, N, generator=g, dtype=f64) < outlier_frac
    noise = torch.where(outl[..., None], torch.randn(B, N, 2, generator=g, dtype=f64) * 20, noise)
    u = proj + noise
    inv_std = torch.rand(B, N, 2, generator=g, dtype=f64) + 0.5
    bbox = bbox3d_from_scale(ext).expand(B, 8, 3).contiguous()
    # PnP start: gt o (rotvec N(0,0.08^2) rad, t*(1+N(0,0.03^2)))
    rv = torch.randn(B, 3, generator=g, dtype=f64) * 0.08
    ang = rv.norm(dim=-1, keepdim=True)
    dq = torch.cat(((ang / 2).cos(), rv / ang * (ang / 2).sin()), -1)
    q0 = _quat_mul(q, dq)
    t0 = t * (1 + torch.randn(B, 3, generator=g, dtype=f64) * 0.03)
    out = dict(K=K, pose=torch.cat((q, t), -1), pts3d=X, pts2d=u, inv_std=inv_std, bbox_3d=bbox,
               start=torch.cat((q0, t0), -1))
    return {k: v.to(dtype).contiguous() for k, v in out.items()}


def make_head_logits(B: int, S: int, H: int, W: int, seed: int = 0, dtype=torch.float32, bump: float = 8.0,
                     sigma: float = 2.0):
    """Keypoint-head logits: N(0,1) + bump * Gaussian(sigma px) at a random in-frame location."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, S, H, W, generator=g, dtype=torch.float32)
    cx = torch.rand(B, S, 1, 1, generator=g) * (W - 9) + 4
    cy = torch.rand(B, S, 1, 1, generator=g) * (H - 9) + 4
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    logits += bump * torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma * sigma))
    return logits.to(dtype).contiguous()
