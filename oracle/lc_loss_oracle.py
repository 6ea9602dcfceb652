"""CPU oracle for the linear-covariance pose loss (TEST INFRASTRUCTURE ONLY).

This file is a closed-form restatement, in plain torch-CPU ops (fp32 or fp64), of the
reference's `lib/cov_mixed.py:100-150 Loss_cov_mixed` and everything it calls
(`lib/nll/pnp_auto.py`, `lib/nll/pnp_utils.py`, `lib/transforms/*`).  It is the
checker for the HIP kernels in `lc_amd/csrc/` -- it is never the product path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.

Parity is PINNED: `tests/golden/gen_golden.py` imports the unmodified reference in the
build container and stores inputs + loss + gradients + intermediates (fp32 and fp64) in
`tests/golden/lc_loss_*.npz`; `tests/test_oracle_loss.py` checks this file against them.

Gradients come from torch autograd over the closed form (independent of the hand-derived
analytic backward the HIP kernel implements, which makes it a real cross-check).
"""
from __future__ import annotations

import torch
from torch import Tensor


def quaternion_to_matrix(q: Tensor):
    """`lib/transforms/rotation_conversions.py:39-68`.

    NOTE two_s = 2/||q|| (not 2/||q||^2) -- reproduced on purpose (line 52).
    Returns (R_sic, R_true, rho): the matrix the reference uses, the proper rotation
    of the normalised quaternion, and rho=||q||.  For unit q all three agree.
    """
    r, i, j, k = torch.unbind(q, -1)
    rho = torch.linalg.vector_norm(q, dim=-1)

    def build(two_s):
        o = torch.stack(
            (
                1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
            ),
            -1,
        )
        return o.reshape(q.shape[:-1] + (3, 3))

    return build(2.0 / rho), build(2.0 / (rho * rho)), rho


def skew(v: Tensor) -> Tensor:
    """[v]x, `lib/nll/pnp_utils.py:36-51 make_skew_symm`."""
    z = torch.zeros_like(v[..., 0])
    a, b, c = v.unbind(-1)
    return torch.stack((z, -c, b, c, z, -a, -b, a, z), -1).reshape(v.shape[:-1] + (3, 3))


def twice_huber(v: Tensor, delta) -> Tensor:
    """`lib/cov_mixed.py:10-13`."""
    delta = delta.detach() if isinstance(delta, Tensor) else delta
    return torch.where(v > delta, delta * (2 * v - delta), v ** 2)


def project_apply(K: Tensor, X: Tensor, R: Tensor, t: Tensor, min_z: float = 0.1) -> Tensor:
    """`lib/transforms/transforms.py:47-63` (full 3x3 K, z clamped at 0.1)."""
    Xc = X @ R.mT + t[..., None, :]
    xf = Xc @ K.mT
    z = xf[..., 2:3].clamp(min=min_z)
    return xf[..., :2] / z


def reproj_jacobian(K: Tensor, R: Tensor, t: Tensor, X: Tensor, meas: Tensor):
    """Residual r, 2x6 Jacobian J and per-coordinate 6x6 Hessian of r at delta=0.

    Closed form of `lib/nll/pnp_auto.py:13-56 residual_with_jac6d` (right perturbation
    R*exp(w), t+tau; camera-frame point NOT z-clamped) and of the `functorch.jacfwd`
    of it in `pnp_auto.py:59-83` (which equals the exact second derivative of r under the
    exponential map because `pnp_utils.py:54-78` is 2nd-order consistent at 0).
    Shapes: K (B,3,3) R (B,3,3) t (B,3) X (B,N,3) meas (B,N,2) -> r (B,N,2), J (B,N,2,6), Hr (B,N,2,6,6)
    """
    B, N = X.shape[:2]
    Xc = X @ R.mT + t[:, None, :]
    inv_z = 1.0 / Xc[..., 2]
    uv0 = Xc[..., :2] * inv_z[..., None]
    eye2 = torch.eye(2, dtype=X.dtype).expand(B, N, 2, 2)
    P = inv_z[..., None, None] * torch.cat((eye2, -uv0[..., None]), -1)  # (B,N,2,3)
    M0 = -(R[:, None] @ skew(X))  # (B,N,3,3)
    eye3 = torch.eye(3, dtype=X.dtype).expand(B, N, 3, 3)
    T = torch.cat((M0, eye3), -1)  # (B,N,3,6)
    K2 = K[:, None, :2, :2]
    J = K2 @ (P @ T)  # (B,N,2,6)
    r = (uv0[..., None, :] @ K2.mT)[..., 0, :] + K[:, None, :2, 2] - meas

    # second derivatives of uv0_a=(x/z, y/z) w.r.t. the camera-frame point
    x, y, z = Xc.unbind(-1)
    iz2, iz3 = inv_z ** 2, inv_z ** 3
    Q = X.new_zeros(B, N, 2, 3, 3)
    Q[..., 0, 0, 2] = -iz2
    Q[..., 0, 2, 0] = -iz2
    Q[..., 0, 2, 2] = 2 * x * iz3
    Q[..., 1, 1, 2] = -iz2
    Q[..., 1, 2, 1] = -iz2
    Q[..., 1, 2, 2] = 2 * y * iz3
    H0 = torch.einsum('bnpi,bnapq,bnql->bnail', T, Q, T)
    # second derivative of exp([w]x)X at 0: -X d_il + (e_i X_l + e_l X_i)/2
    e = torch.eye(3, dtype=X.dtype)
    S = (-X[..., None, None, :] * e[:, :, None]
         + 0.5 * (e[:, None, :] * X[..., None, :, None] + e[None, :, :] * X[..., :, None, None]))  # (B,N,3,3,3) [i,l,:]
    RS = torch.einsum('bdk,bnilk->bnild', R, S)
    H0[..., :3, :3] = H0[..., :3, :3] + torch.einsum('bnad,bnild->bnail', P, RS)
    Hr = torch.einsum('bca,bnail->bncil', K[:, :2, :2], H0)
    return r, J, Hr


def bbox_jacobian(R_true: Tensor, rho: Tensor, bbox: Tensor) -> Tensor:
    """`lib/cov_mixed.py:42-65 jac_update2alter` with `xform_3d` (:73-75): d(R b_k + t)/d(delta) at 0
    = [ -rho R_true [b_k]x | I3 ], stacked to (B,24,6)."""
    B = bbox.shape[0]
    rot = -(rho[:, None, None, None] * (R_true[:, None] @ skew(bbox)))  # (B,8,3,3)
    eye3 = torch.eye(3, dtype=bbox.dtype).expand(B, 8, 3, 3)
    return torch.cat((rot, eye3), -1).reshape(B, 24, 6)


def loss_cov_3d(diag: Tensor, dim: int = 3) -> Tensor:
    """`lib/cov_mixed.py:83-89` (dim=3) and `loss_cov_2d` `:91-97` (dim=2)."""
    B = len(diag)
    good = (diag > 0).all(dim=-1, keepdim=True)
    pw = diag.reshape(B, -1, dim)
    return torch.where(good, pw.sum(-1), 1).sqrt().mean(-1)


def bbox_jacobian_2d(K: Tensor, R: Tensor, t: Tensor, R_true: Tensor, rho: Tensor, bbox: Tensor) -> Tensor:
    """`jac_update2alter` with `xform_2d` (cov_mixed.py:78-80): d(project_apply(K, R b + t))/d(delta) at 0, (B,16,6)."""
    B = bbox.shape[0]
    G3 = bbox_jacobian(R_true, rho, bbox).reshape(B, 8, 3, 6)
    Xc = bbox @ R.mT + t[:, None, :]
    xf = Xc @ K.mT
    zpass = (xf[..., 2:3] >= 0.1).to(K.dtype)
    zc = xf[..., 2:3].clamp(min=0.1)
    proj = xf[..., :2] / zc
    Pa = (K[:, None, :2, :] - zpass[..., None] * proj[..., None] * K[:, None, 2:3, :]) / zc[..., None]  # (B,8,2,3)
    return (Pa @ G3).reshape(B, 16, 6)


def loss_cov_mixed(K: Tensor, pose: Tensor, pts3d: Tensor, pts2d: Tensor, inv_std: Tensor,
                   valid: Tensor | None, *, bbox_3d: Tensor, max_err_len=32, rel_thresh=3, w_e_thresh=4, cov_2d=False,
                   return_intermediates: bool = False):
    """Closed form of `lib/cov_mixed.py:100-150` (cov_2d=False is the branch every caller uses)."""
    R, R_true, rho = quaternion_to_matrix(pose[..., :4])
    t = pose[..., 4:7]
    proj = project_apply(K, pts3d, R, t)
    err = pts2d - proj
    # clamp_error (:16-24): no-grad rescale, identity gradient
    with torch.no_grad():
        ln = torch.linalg.vector_norm(err, dim=-1) + 1e-6
        f = ((ln - max_err_len) / ln).unsqueeze(-1)
        delta = f * err * (f > 0)
    e = err - delta
    # robust_weights_cov (:27-39)
    a = e.abs()
    with torch.no_grad():
        if valid is not None:
            vm, vc = valid.unsqueeze(-1), valid.sum(-1, keepdim=True)
            mean_abs = (a * vm).sum(-2) / vc
        else:
            mean_abs = a.mean(-2)
    c = twice_huber(a, mean_abs.unsqueeze(-2) * rel_thresh)
    with torch.no_grad():
        w_e = inv_std ** 2 * c
        mean_w_e = (w_e * vm).sum(-2) / vc if valid is not None else w_e.mean(-2)
        d_s = torch.sqrt(mean_w_e.unsqueeze(-2) * w_e_thresh / (c + 1e-6))
    w = twice_huber(inv_std, d_s)

    # pnp_auto.weighted_pnp_jac_wrt_pts2d (:111-135) on detached geometry; measurements := proj.detach()
    r, J, Hr = reproj_jacobian(K.detach(), R.detach(), t.detach(), pts3d.detach(), proj.detach())
    Hfull = torch.einsum('bnc,bncil->bil', w, J[..., :, None] * J[..., None, :] + r[..., None, None] * Hr)
    Hfull = 0.5 * Hfull + 0.5 * Hfull.mT  # make_sure_symmetric (pnp_utils.py:134-137)
    info = torch.linalg.cholesky_ex(Hfull.detach())[1]  # make_sure_SPD (:140-157)
    eye6 = torch.eye(6, dtype=Hfull.dtype)
    Hfix = torch.where((info != 0)[:, None, None], eye6, Hfull)
    L = torch.linalg.cholesky_ex(Hfix)[0]
    S = torch.cholesky_inverse(L)  # prior_update_cov (pnp_auto.py:107)
    A = torch.einsum('bkj,bncj,bnc->bknc', S, J, w).flatten(-2)  # (B,6,2N)  = H^-1 J^T W

    gdim = 2 if cov_2d else 3
    if cov_2d:
        G = bbox_jacobian_2d(K.detach(), R.detach(), t.detach(), R_true.detach(), rho.detach(), bbox_3d)
    else:
        G = bbox_jacobian(R_true.detach(), rho.detach(), bbox_3d)
    prior_diag = ((G @ S) * G).sum(-1)  # transformed_cov_from_jac (:68-70)
    prior_error = loss_cov_3d(prior_diag, gdim)
    cc = c.flatten(-2)
    half = (A * cc.unsqueeze(-2)) @ A.mT * 0.5
    U = half + half.mT
    cov_diag = ((G @ U) * G).sum(-1)
    cov_err = loss_cov_3d(cov_diag, gdim)
    dlt = (G @ (A @ e.detach().flatten(-2).unsqueeze(-1))).squeeze(-1)
    lin = torch.linalg.vector_norm(dlt.reshape(dlt.shape[:-1] + (8, gdim)), dim=-1).mean(-1)
    loss = prior_error.log() + 0.5 * (cov_err + lin) / prior_error
    if return_intermediates:
        return loss, dict(w=w, c=c, Hinv=S, A=A, G=G, e=e, info=info,
                          prior_error=prior_error, cov_err=cov_err, linear_err=lin)
    return loss


def loss_and_grads(K, pose, pts3d, pts2d, inv_std, valid, bbox_3d, grad_out=None, want_pts3d=True, **kw):
    """Convenience for tests/bench: loss (B,), d/d pts2d, d/d inv_std, d/d pts3d for cotangent grad_out (default ones)."""
    pts2d = pts2d.detach().clone().requires_grad_(True)
    inv_std = inv_std.detach().clone().requires_grad_(True)
    pts3d = pts3d.detach().clone().requires_grad_(want_pts3d)
    loss = loss_cov_mixed(K, pose, pts3d, pts2d, inv_std, valid, bbox_3d=bbox_3d, **kw)
    go = torch.ones_like(loss) if grad_out is None else grad_out
    ins = [pts2d, inv_std] + ([pts3d] if want_pts3d else [])
    gs = torch.autograd.grad(loss, ins, go, allow_unused=True)
    return loss.detach(), gs[0], gs[1], (gs[2] if want_pts3d else None)
