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


# ---- synthetic inputs of Loss_fn.forward / solve_pnp (gt_dict, out_dict); tests/golden/gen_golden_lossfn.py feeds the same ones to the
# reference class, so the committed lossfn_* trajectories are functions of exactly these tensors ----

def sparse_inputs(B=6, N=16, seed=0):
    b = make_batch(B, N, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    gt = dict(pose_best=b["pose"], out_K=b["K"], pts3d=b["pts3d"], bbox_3d=b["bbox_3d"],
              msk_noc=torch.ones(B, 4, 4, dtype=torch.bool), msk_vis=torch.ones(B, 4, 4))
    out = dict(pts2d=b["pts2d"], pts2d_std=torch.rand(B, N, 2, generator=g) * 1.5 + 0.5)
    return gt, out


def dense_inputs(B=3, H=16, W=16, seed=0):
    """A 16x16 output grid looking at a synthetic surface: xyz head ~ back-projected pixels + noise."""
    g = torch.Generator().manual_seed(seed)
    b = make_batch(B, 4, seed=seed + 7, rotate_K=False)
    K = b["K"].clone()
    K[:, 0, 0] = 110.0
    K[:, 1, 1] = 110.0
    K[:, 0, 2] = W / 2
    K[:, 1, 2] = H / 2
    pose = b["pose"].clone()
    pose[:, 4:6] = 0
    pose[:, 6] = 500.0
    from .transforms import quaternion_rep_to_RT, gen_uv

    R, t = quaternion_rep_to_RT(pose)
    uv = gen_uv((H, W))  # (H,W,2)
    ones = torch.ones(H, W, 1)
    rays = torch.cat((uv, ones), -1).reshape(1, -1, 3) @ torch.linalg.inv(K).mT  # (B,HW,3)
    z = 500.0 + 10 * torch.randn(B, H * W, 1, generator=g)
    Xc = rays * z
    Xm = (Xc - t[:, None]) @ R  # R^T (Xc - t)
    noc_scale = torch.tensor(EXTENT_MM).expand(B, 3).contiguous()
    noc = (Xm / noc_scale[:, None]).mT.reshape(B, 3, H, W)
    xyz_noc = noc + 0.02 * torch.randn(B, 3, H, W, generator=g)
    msk_vis = (torch.rand(B, H, W, generator=g) > 0.3).float()
    gt = dict(pose_best=pose, out_K=K, bbox_3d=b["bbox_3d"], noc_scale=noc_scale, msk_noc=msk_vis > 0, msk_vis=msk_vis,
              xyz_noc_tgt=noc * msk_vis[:, None])
    out = dict(xyz_noc=xyz_noc, xyz_weight_logits=torch.randn(B, 2, H, W, generator=g),
               xyz_weights_scale=torch.exp(torch.randn(B, 1, 1, 1, generator=g) * 0.2 + 3.0),
               msk_vis_logits=torch.randn(B, 1, H, W, generator=g))
    return gt, out


def bin_inputs(B=3, H=16, W=16, seed=0, bits=(6, 6, 5)):
    """ZebraPose structure: binary surface codes instead of the continuous xyz head, with a model transform."""
    from . import floatbits as fb

    gt, out = dense_inputs(B, H, W, seed)
    g = torch.Generator().manual_seed(seed + 50)
    noc = (gt["xyz_noc_tgt"] / 1.0).permute(0, 2, 3, 1).clamp(-0.999, 0.999)  # (B,H,W,3) normalised target coordinates
    mod_bits, raw_bits = fb.nn_noc2target(noc, list(bits))
    C = sum(bits)
    logits = (mod_bits.float() * 2 - 1) * (torch.rand(B, C, H, W, generator=g) * 3 + 0.2)
    logits = torch.where(torch.rand(B, C, H, W, generator=g) < 0.12, -logits, logits)
    ang = 0.3
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, 0, 0] = T[:, 1, 1] = math.cos(ang)
    T[:, 0, 1], T[:, 1, 0] = -math.sin(ang), math.sin(ang)
    T[:, :3, 3] = torch.tensor([1.5, -2.0, 0.5])
    gt.pop("xyz_noc_tgt")
    gt.update(xyz_noc_bin_tgt=mod_bits, xyz_noc_bin_raw=raw_bits, bit_cnt=list(bits), model_transform=T)
    out.pop("xyz_noc")
    out["xyz_noc_bin"] = logits
    return gt, out


# ---- test-time workloads of the reference's own configs (test.py:67-136; configs/zlmo.yaml:30-37, configs/glmo.yaml:28-32) ----

TEST_TIME_CONFIGS = {
    # pnp_solver block of the config + the head's output grid and code layout
    "zlmo": dict(pnp_solver=dict(seg_thresh=0.5, dense_sample=1, rel_reproj_err=True, dense_point_select="quantile_in_mask", quantile=0.2,
                                 solvers=["weighted_filtered"]), H=128, W=128, bits=(7, 7, 7), model_transform=True),
    "glmo": dict(pnp_solver=dict(dense_point_select="quantile", quantile=0.3, solvers=["weighted"]), H=64, W=64, bits=None,
                 model_transform=False),
    # BASELINE configs[0], "reference plumbing": glmo on 128x128 crops => 32x32 maps (the stride-4 decoder), stride 2 => N = 256 correspondences
    "plumb": dict(pnp_solver=dict(dense_point_select="quantile", quantile=0.3, solvers=["weighted"]), H=32, W=32, bits=None,
                  model_transform=False),
}


def test_time_inputs(name: str, B: int = 64, seed: int = 0, flip: float = 0.02, train: bool = False):
    """Synthetic network outputs of the dense heads at test time, shaped like the named config's: every object is an ellipsoid with the
    LM-O extents seen under a random pose, rendered by ray casting (so pixel <-> model point correspondences are exact up to the noise
    added below and the visible region is a blob of ~20-30 % of the crop); zlmo: 21 code planes (Gray code of the
    model-transformed, scaled coordinates, `flip` of the bits wrong: gross outliers for the RANSAC), glmo: the continuous xyz head.
    Weight logits are higher on the object, visibility logits follow the silhouette with a few errors at its rim.
    -> (cfg dict for AttrDict, gt_dict, out_dict) of CPU tensors; gt_dict['pose_best'] is the pose to recover.
    `train=True` adds what `Loss_fn.forward` reads at training time (`losses.py:49-67,132-139`: `msk_noc`, `xyz_noc_tgt` or the code
    targets `xyz_noc_bin_tgt / _raw`), an occluder over part of `msk_vis`, and gross errors on a few pixels of the xyz head; every extra
    random draw comes after the test-time ones, so the test-time tensors of a (name, B, seed) do not depend on the flag."""
    from . import floatbits as fb

    spec = TEST_TIME_CONFIGS[name]
    H, W, bits = spec["H"], spec["W"], spec["bits"]
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    q = torch.randn(B, 4, generator=g, dtype=f64)
    q = q / q.norm(dim=-1, keepdim=True)
    q = torch.where(q[:, :1] < 0, -q, q)
    z = torch.rand(B, generator=g, dtype=f64) * 400 + 500
    ext = torch.tensor(EXTENT_MM, dtype=f64)
    f = (torch.rand(B, generator=g, dtype=f64) * 0.15 + 0.55) * W * z / (2 * ext.max())  # the object spans 55-70 % of the crop's width (zoomed crops, dataset.py:402-423)
    th = torch.rand(B, generator=g, dtype=f64) * (2 * math.pi)
    K = torch.zeros(B, 3, 3, dtype=f64)
    K[:, 0, 0], K[:, 0, 1], K[:, 1, 0], K[:, 1, 1] = f * th.cos(), -f * th.sin(), f * th.sin(), f * th.cos()
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = W / 2, H / 2, 1
    t = torch.stack(((torch.rand(B, generator=g, dtype=f64) - 0.5) * 0.15 * W * z / f, (torch.rand(B, generator=g, dtype=f64) - 0.5) * 0.15 * H * z / f, z), -1)
    R = _quat_to_R(q)
    # ray casting in the model frame: o + s d on the ellipsoid |x / ext| = 1, nearest hit
    ys, xs = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    pix = torch.stack((xs, ys, torch.ones_like(xs)), -1).reshape(1, H * W, 3)
    d = (pix @ torch.linalg.inv(K).mT) @ R            # R^T K^-1 (u, v, 1)
    o = -(t[:, None, :] @ R)                          # -R^T t
    dn, on = d / ext, o / ext
    a, bq, c = (dn * dn).sum(-1), 2 * (dn * on).sum(-1), (on * on).sum(-1) - 1
    disc = bq * bq - 4 * a * c
    hit = disc > 0
    s = (-bq - disc.clamp_min(0).sqrt()) / (2 * a)
    Xm = torch.where(hit[..., None], o + s[..., None] * d, torch.zeros((), dtype=f64))  # (B,HW,3) model points; background: 0
    hit = hit.reshape(B, H, W)
    if spec["model_transform"]:
        ang = torch.rand(B, generator=g, dtype=f64) * 0.6 - 0.3
        T = torch.eye(4, dtype=f64).repeat(B, 1, 1)
        T[:, 0, 0] = T[:, 1, 1] = ang.cos()
        T[:, 0, 1], T[:, 1, 0] = -ang.sin(), ang.sin()
        T[:, :3, 3] = torch.randn(B, 3, generator=g, dtype=f64) * 2
        Xt = Xm @ T[:, :3, :3].mT + T[:, None, :3, 3]
        noc_scale = (ext * 1.25 + 6).expand(B, 3).contiguous()  # the transformed coordinates stay inside (-1, 1)
    else:
        T, Xt, noc_scale = None, Xm, ext.expand(B, 3).contiguous()
    noc = (Xt / noc_scale[:, None]).reshape(B, H, W, 3)
    msk_vis = hit.float()
    wl = torch.randn(B, 2, H, W, generator=g) + 6 * msk_vis[:, None]  # background weights e^-6 of the object's
    out = dict(xyz_weight_logits=wl, xyz_weights_scale=torch.exp(torch.randn(B, 1, 1, 1, generator=g) * 0.2 + (3.0 if bits is None else 4.5)),
               msk_vis_logits=(msk_vis[:, None] * 2 - 1) * 4 + torch.randn(B, 1, H, W, generator=g) * 2)
    gt = dict(pose_best=torch.cat((q, t), -1).float(), out_K=K.float(), noc_scale=noc_scale.float(), msk_vis=msk_vis,
              bbox_3d=bbox3d_from_scale(ext).expand(B, 8, 3).contiguous().float(),
              out_pix_scale=(torch.rand(B, generator=g) * 1.5 + 0.5))
    if bits is None:
        out["xyz_noc"] = (noc.permute(0, 3, 1, 2) + 0.01 * torch.randn(B, 3, H, W, generator=g, dtype=f64)).float().contiguous()
    else:
        mod_bits, _raw = fb.nn_noc2target(noc.float().clamp(-0.999, 0.999), list(bits))
        C = sum(bits)
        logits = (mod_bits.float() * 2 - 1) * (torch.rand(B, C, H, W, generator=g) * 3 + 0.2)
        out["xyz_noc_bin"] = torch.where(torch.rand(B, C, H, W, generator=g) < flip, -logits, logits).contiguous()
        gt.update(bit_cnt=list(bits), model_transform=T.float())
    if train:
        # an occluder: a rectangle of the crop is not visible (msk_vis is a subset of msk_noc, dataset.py:453-458); the heads do not know it
        x0, y0 = (torch.rand(B, generator=g) * 0.6 * W).long(), (torch.rand(B, generator=g) * 0.6 * H).long()
        x1, y1 = x0 + (torch.rand(B, generator=g) * 0.4 * W).long() + 2, y0 + (torch.rand(B, generator=g) * 0.4 * H).long() + 2
        ysl, xsl = torch.arange(H)[None, :, None], torch.arange(W)[None, None, :]
        occ = (xsl >= x0[:, None, None]) & (xsl < x1[:, None, None]) & (ysl >= y0[:, None, None]) & (ysl < y1[:, None, None])
        gt["msk_noc"] = hit
        gt["msk_vis"] = (hit & ~occ).float()
        if bits is None:
            gt["xyz_noc_tgt"] = (noc.permute(0, 3, 1, 2) * hit[:, None]).float().contiguous()
            gross = (torch.rand(B, 1, H, W, generator=g) < 0.03).float() * 0.3
            out["xyz_noc"] = (out["xyz_noc"] + gross * torch.randn(B, 3, H, W, generator=g)).contiguous()
        else:
            # the label of a pixel off the object is the code of the coordinate 0 (losses.py:58-62: the transformed points are masked)
            tgt = ((Xt * hit.reshape(B, H * W, 1)) / noc_scale[:, None]).reshape(B, H, W, 3).float()
            gt["xyz_noc_bin_tgt"], gt["xyz_noc_bin_raw"] = fb.nn_noc2target(tgt, list(bits))
    return dict(spec["pnp_solver"]), gt, out


# ---- Loss_fn.forward at the reference's own training shapes (configs/glmo.yaml:9,63-65,72-79, configs/zlmo.yaml:74-83,
# configs/gsplmo.yaml loss block); tests/golden/gen_golden_lossfn.py runs the reference class on exactly these tensors ----

TRAIN_LOSS_CONFIGS = {
    # loss block of configs/glmo.yaml:72-79 (dense_sample defaults to 2 => N = 32 x 32 = 1024 of the 64x64 maps)
    "dense_glmo": dict(pose_loss_cfg=dict(clip_weight_grad=True), pose_loss_start_step=2000, pose_loss_start_epoch=1, w_loss_pose=0.02,
                       w_loss_seg=0.25, w_loss_noc=1),
    # loss block of configs/zlmo.yaml:74-83 (128x128 maps, dense_sample 3 => N = 43 x 43 = 1849; 7+7+7 code planes)
    "bin_zlmo": dict(pose_loss_cfg=dict(dense_sample=3, clip_weight_grad=True), seg_loss_type="L1", pose_loss_start_step=3000,
                     pose_loss_start_epoch=0, w_loss_pose=0.03, w_loss_noc_bin=3, w_loss_seg=1),
    # BASELINE configs[0] (16 crops of 128x128 through glmo: 32x32 maps, stride 2 => N = 256, the one-workgroup kernel's largest size): glmo's block
    "dense_plumb": dict(pose_loss_cfg=dict(clip_weight_grad=True), pose_loss_start_step=2000, pose_loss_start_epoch=1, w_loss_pose=0.02,
                        w_loss_seg=0.25, w_loss_noc=1),
    # loss block of configs/gsplmo.yaml at BASELINE's B=256, N=64 keypoints
    "sparse_metric": dict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True), pose_loss_start_step=4000, pose_loss_start_epoch=1,
                          w_loss_kpts=1, w_loss_pose=0.7),
}


def train_inputs(kind: str, seed: int = 0, B: int = None):
    """(gt_dict, out_dict) of `Loss_fn.forward` for one of TRAIN_LOSS_CONFIGS' kinds, CPU tensors."""
    if kind == "sparse_metric":
        return sparse_inputs(B=B or 256, N=64, seed=seed)
    name = {"dense_glmo": "glmo", "bin_zlmo": "zlmo", "dense_plumb": "plumb"}[kind]
    _cfg, gt, out = test_time_inputs(name, B=B or (16 if kind == "dense_plumb" else 4), seed=seed + 20, flip=0.08, train=True)
    gt.pop("out_pix_scale")
    return gt, out
