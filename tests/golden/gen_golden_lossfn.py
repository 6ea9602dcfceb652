"""Golden trajectories of the reference's `losses.Loss_fn.forward` (sparse and dense branch), see gen_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from lc_amd import synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402


def sparse_inputs(B=6, N=16, seed=0):
    b = synth.make_batch(B, N, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    gt = dict(pose_best=b["pose"], out_K=b["K"], pts3d=b["pts3d"], bbox_3d=b["bbox_3d"],
              msk_noc=torch.ones(B, 4, 4, dtype=torch.bool), msk_vis=torch.ones(B, 4, 4))
    out = dict(pts2d=b["pts2d"], pts2d_std=torch.rand(B, N, 2, generator=g) * 1.5 + 0.5)
    return gt, out


def dense_inputs(B=3, H=16, W=16, seed=0):
    """A 16x16 output grid looking at a synthetic surface: xyz head ~ back-projected pixels + noise."""
    g = torch.Generator().manual_seed(seed)
    b = synth.make_batch(B, 4, seed=seed + 7, rotate_K=False)
    K = b["K"].clone()
    K[:, 0, 0] = 110.0
    K[:, 1, 1] = 110.0
    K[:, 0, 2] = W / 2
    K[:, 1, 2] = H / 2
    pose = b["pose"].clone()
    pose[:, 4:6] = 0
    pose[:, 6] = 500.0
    from lc_amd.transforms import quaternion_rep_to_RT, gen_uv

    R, t = quaternion_rep_to_RT(pose)
    uv = gen_uv((H, W))  # (H,W,2)
    ones = torch.ones(H, W, 1)
    rays = torch.cat((uv, ones), -1).reshape(1, -1, 3) @ torch.linalg.inv(K).mT  # (B,HW,3)
    z = 500.0 + 10 * torch.randn(B, H * W, 1, generator=g)
    Xc = rays * z
    Xm = (Xc - t[:, None]) @ R  # R^T (Xc - t)
    noc_scale = torch.tensor(synth.EXTENT_MM).expand(B, 3).contiguous()
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
    from lc_amd import floatbits as fb

    gt, out = dense_inputs(B, H, W, seed)
    g = torch.Generator().manual_seed(seed + 50)
    noc = (gt["xyz_noc_tgt"] / 1.0).permute(0, 2, 3, 1).clamp(-0.999, 0.999)  # (B,H,W,3) normalised target coordinates
    mod_bits, raw_bits = fb.nn_noc2target(noc, list(bits))
    C = sum(bits)
    logits = (mod_bits.float() * 2 - 1) * (torch.rand(B, C, H, W, generator=g) * 3 + 0.2)
    logits = torch.where(torch.rand(B, C, H, W, generator=g) < 0.12, -logits, logits)
    ang = 0.3
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, 0, 0] = T[:, 1, 1] = float(np.cos(ang))
    T[:, 0, 1], T[:, 1, 0] = -float(np.sin(ang)), float(np.sin(ang))
    T[:, :3, 3] = torch.tensor([1.5, -2.0, 0.5])
    gt.pop("xyz_noc_tgt")
    gt.update(xyz_noc_bin_tgt=mod_bits, xyz_noc_bin_raw=raw_bits, bit_cnt=list(bits), model_transform=T)
    out.pop("xyz_noc")
    out["xyz_noc_bin"] = logits
    return gt, out


BIN_CFG = dict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True, dense_sample=2, max_err_len=32), pose_loss_start_step=2,
               pose_loss_start_epoch=0, loss_pose_nz_step=0, w_loss_noc_bin=1, w_loss_seg=1, w_loss_pose=0.05, seg_loss_type="L1")
SPARSE_CFG = dict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True), pose_loss_start_step=6, pose_loss_start_epoch=0,
                  loss_pose_nz_step=2, w_loss_kpts=1, w_loss_pose=0.7)
DENSE_CFG = dict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True, clip_scale_grad=True, clip_pts_grad=True, dense_sample=2,
                                    max_err_len=32), pose_loss_start_step=3, pose_loss_start_epoch=0, loss_pose_nz_step=0,
                 w_loss_noc=1, w_loss_seg=1, w_loss_pose=0.05, seg_loss_type="L1")


def run(Loss_fn_cls, kind, steps, dtype=torch.float32, device=None):
    """Shared by the generator (reference class) and the tests (lc_amd class): returns a flat record of the trajectory."""
    cfg = AttrDict({"sparse": SPARSE_CFG, "dense": DENSE_CFG, "bin": BIN_CFG}[kind])
    fn = Loss_fn_cls(cfg, AttrDict(), 17 if kind == "bin" else 0)
    if device is not None:
        fn = fn.to(device)
    rec = {}
    for i, step in enumerate(steps):
        gt, out = {"sparse": sparse_inputs, "dense": dense_inputs, "bin": bin_inputs}[kind](seed=i)
        gt = {k: ((v.to(dtype) if v.is_floating_point() else v).to(device) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
        leaves = {k: v.to(dtype).to(device).clone().requires_grad_(True) for k, v in out.items()}
        np.random.seed(1000 + i)  # random sub-sampling phase (losses.py:152)
        loss_dict, w_loss_dict = fn(gt, leaves, 0, step, 10)
        total = sum(w_loss_dict.values())
        grads = torch.autograd.grad(total, list(leaves.values()), allow_unused=True)
        for k, v in loss_dict.items():
            rec[f"s{i}_loss_{k}"] = v.detach().double().cpu().numpy()
        for k, v in w_loss_dict.items():
            rec[f"s{i}_wloss_{k}"] = v.detach().double().cpu().numpy()
        for k, gk in zip(leaves, grads):
            if gk is not None:
                rec[f"s{i}_grad_{k}"] = gk.double().cpu().numpy()
        for k, v in fn.state_dict().items():
            rec[f"s{i}_state_{k}"] = v.detach().double().cpu().numpy().copy()  # copy: the state may be updated in place
    rec["steps"] = np.asarray(steps)
    return rec


def gen_lossfn():
    sys.path.insert(0, os.environ.get("LC_REFERENCE", "/root/reference"))
    import losses as ref_losses

    for kind, steps in (("sparse", [0, 2, 4, 6, 9]), ("dense", [0, 1, 2, 5]), ("bin", [0, 1, 3])):
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            rec = run(ref_losses.Loss_fn, kind, steps, dt)
            path = os.path.join(HERE, f"lossfn_{kind}_{tag}.npz")
            np.savez_compressed(path, **rec)
            print(kind, tag, {k: float(v) for k, v in rec.items() if k.startswith("s0_loss_")}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    gen_lossfn()
