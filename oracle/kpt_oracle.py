"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the sparse heads' keypoint NLL, `Loss_fn.sparse_kpt_loss` (losses.py:318-326):
  R, t = quaternion_rep_to_RT(pose)        transforms.py:6-32, rotation_conversions.py:39-68 (two_s = 2/|q|, sic)
  proj = project_apply(K, X, R, t)         transforms.py:47-63 (z of K*Xc clamped at 0.1)
  nll  = mean(log std + |u - proj| / std)
Pinned through the `Loss_fn` goldens (tests/golden/lossfn_sparse_*.npz hold the reference's loss_kpts and its gradients).
"""
import torch


def _rotation(q):
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / q.norm(dim=-1)
    return torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1).reshape(q.shape[:-1] + (3, 3))


def project(K, pose, X):
    R, t = _rotation(pose[..., :4]), pose[..., 4:]
    xf = (X @ R.mT + t[..., None, :]) @ K.mT
    return xf[..., :2] / xf[..., 2:3].clamp(min=0.1)


def kpt_nll_per_sample(K, pose, X, u, std):
    return (torch.log(std) + (u - project(K, pose, X)).abs() / std).sum(dim=(-1, -2))


def nll_and_grads(K, pose, X, u, std):
    """-> per-sample sums (B,), d/du and d/dstd of each sample's own sum (unit cotangent), in the inputs' dtype."""
    u = u.detach().clone().requires_grad_(True)
    std = std.detach().clone().requires_grad_(True)
    nll = kpt_nll_per_sample(K, pose, X, u, std)
    du, ds = torch.autograd.grad(nll.sum(), (u, std))
    return nll.detach(), du, ds
