"""Pose / projection helpers with the reference's exact conventions (`lib/transforms/transforms.py`,
`lib/transforms/rotation_conversions.py:39-68`).  Cheap torch glue used by the loss surface, not a hot op."""
from __future__ import annotations

import torch
from torch import Tensor


def quaternion_to_matrix(quaternions: Tensor) -> Tensor:
    """wxyz -> R with the reference's two_s = 2/||q|| (sic, rotation_conversions.py:52)."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / torch.linalg.vector_norm(quaternions, dim=-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def quaternion_rep_to_RT(quaternion_reps: Tensor):
    """(*,7) w,x,y,z,tx,ty,tz -> R (*,3,3), t (*,3)   (transforms.py:6-32)"""
    return quaternion_to_matrix(quaternion_reps[..., :4]), quaternion_reps[..., 4:7]


def matrix_to_quaternion(R: Tensor) -> Tensor:
    """Rotation matrices (*,3,3) -> unit quaternions wxyz (*,4): the row of the symmetric matrix 4 q q^T with the largest
    diagonal entry, divided by twice that component (the component picked is positive), as `rotation_conversions.py:100-159`."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    rows = (
        (1 + m00 + m11 + m22, m21 - m12, m02 - m20, m10 - m01),
        (m21 - m12, 1 + m00 - m11 - m22, m10 + m01, m02 + m20),
        (m02 - m20, m10 + m01, 1 - m00 + m11 - m22, m12 + m21),
        (m10 - m01, m20 + m02, m21 + m12, 1 - m00 - m11 + m22),
    )
    Q = torch.stack([torch.stack(r, -1) for r in rows], -2)                    # (*,4,4) = 4 q q^T
    diag = torch.diagonal(Q, dim1=-2, dim2=-1).clamp_min(0)
    k = diag.argmax(-1, keepdim=True)                                          # best-conditioned component
    row = Q.gather(-2, k[..., None].expand(k.shape + (4,))).squeeze(-2)
    return row / (2 * diag.gather(-1, k).sqrt())


def RT_to_quaternion_rep(Rs: Tensor, ts: Tensor) -> Tensor:
    """R (*,3,3), t (*,3) -> (*,7) w,x,y,z,tx,ty,tz   (transforms.py:35-45)"""
    return torch.cat((matrix_to_quaternion(Rs), ts), dim=-1)


def project_apply(cam_K: Tensor, pts_3d: Tensor, R: Tensor = None, t: Tensor = None, min_z: float = 0.1) -> Tensor:
    """Pinhole projection with the full 3x3 K and z clamped at min_z (transforms.py:47-63)."""
    if R is not None:
        pts_3d = pts_3d @ R.mT + t.squeeze(-1)[..., None, :]
    xformed = pts_3d @ cam_K.transpose(-1, -2)
    z = xformed[..., 2:3].clamp(min=min_z)
    return xformed[..., :2] / z


def gen_uv(shape_hw, device=None, dtype=None) -> Tensor:
    """(H, W, 2) pixel grid, (x, y) order (transforms.py:66-74)."""
    H, W = shape_hw[-2:]
    xs = torch.arange(0, W - 0.5, device=device, dtype=dtype)
    ys = torch.arange(0, H - 0.5, device=device, dtype=dtype)
    x, y = torch.meshgrid((xs, ys), indexing="xy")
    return torch.stack((x, y), dim=-1)
