"""`losses.py` call surface of the reference on the HIP hot path.

Keeps the names `train.py:31,58-59` / `test.py:12` use -- `Loss_fn`, `dense_pnp_matching_from_xyz`, `nn_out_to_xyz` --
with the same arguments, return values and buffer names (`weight_grad_clipper.max_norm`), so a checkpointed
`model.loss_fn` loads strictly.  The pose term goes through `lc_amd.cov_mixed.Loss_cov_mixed`
(one fused HIP launch), the Laplace keypoint NLL through `lc_amd.kpt` (one launch), the weight softmax + strided sub-sampling
through `lc_amd.dense` (one launch each way); what remains of `losses.py:261-386` here is the warm-up blending and the dicts.

The ZebraPose binary-code branch (`xyz_noc_bin`, `losses.py:163-184,196-216`) decodes through `lc_amd.floatbits`
(HIP kernels).

Label preparation (`annots_on_the_fly`, `selete_best_pose`, `xyz_from_homo_z`: `losses.py:68-139`, called at
`train.py:58,116`) is OUTSIDE the hot path and is not rebuilt: the three names exist here as
pass-throughs to the reference's own `losses` module (found through `LC_REFERENCE` or an already imported `losses`), so
`import lc_amd.losses as losses` runs `train.py` wherever the reference checkout is present and raises a clear
ImportError where it is not.
"""
from __future__ import annotations

from operator import itemgetter

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib, dense_aux, floatbits
from . import transforms as xforms
from .cov_mixed import Loss_cov_mixed
from .dense import dense_front_end
from .grad import NormClipper
from .kpt import kpt_nll_mean

# A/B switch for profiles only (scripts/bench_xyz_bin_routes.py): LC_AMD_XYZ_BIN_TORCH=1 sends HIP maps through the torch formulas of
# Loss_xyz_bin, the route a sharded job took up to round 5
import os as _os

_TORCH_SHARDED_ROUTE = _os.environ.get("LC_AMD_XYZ_BIN_TORCH") == "1"


def _reference_losses():
    """The reference's own `losses` module, for the label-preparation names this package does not rebuild."""
    import importlib
    import os
    import sys

    mod = sys.modules.get("losses")
    if mod is not None and getattr(mod, "__file__", None) != __file__ and hasattr(mod, "annots_on_the_fly"):
        return mod
    ref = os.environ.get("LC_REFERENCE")
    if ref and os.path.exists(os.path.join(ref, "losses.py")):
        if ref not in sys.path:
            sys.path.insert(0, ref)
        return importlib.import_module("losses")
    raise ImportError("lc_amd.losses: label preparation (annots_on_the_fly / selete_best_pose / xyz_from_homo_z) is outside the "
                      "MI355X hot path and lives in the reference's losses.py; set LC_REFERENCE=<fulliu/lc checkout> or use lc_amd.dropin")


def annots_on_the_fly(gt_dict, out_dict, cfg_global, step):
    """`losses.py:121-139` (label preparation; pass-through to the reference module)."""
    return _reference_losses().annots_on_the_fly(gt_dict, out_dict, cfg_global, step)


def selete_best_pose(gt_dict, out_dict, sym_aware_started):
    """`losses.py:68-118` (symmetry-aware pose selection; pass-through to the reference module)."""
    return _reference_losses().selete_best_pose(gt_dict, out_dict, sym_aware_started)


def xyz_from_homo_z(*args, **kwargs):
    """Pass-through to the reference's `losses.xyz_from_homo_z`."""
    return _reference_losses().xyz_from_homo_z(*args, **kwargs)


def nn_out_to_xyz(nn_out: Tensor = None, noc_scale_xfd: Tensor = None, *, raw_bits_gt=None, noc_mask=None,
                  model_transform=None, bit_cnt=None, inference=False) -> Tensor:
    """Network output -> (B,H,W,3) object coordinates in mm (`losses.py:17-47`): continuous xyz head (bit_cnt None) or
    binary surface codes decoded with / without the ground-truth bits."""
    if bit_cnt is None:
        assert model_transform is None, "Model transform not implemented for continuous xyz output"
        return nn_out.permute(0, 2, 3, 1) * noc_scale_xfd[:, None, None, :]
    if not inference:
        noc_xformed = floatbits.nn_logits2noc_with_gt(nn_out, raw_bits_gt, bit_cnt, noc_mask)
    else:
        noc_xformed = floatbits.nn_logits2noc(nn_out, bit_cnt)
    xyz_xformed = noc_xformed * noc_scale_xfd[:, None, None, :]
    if model_transform is None:
        return xyz_xformed
    return (xyz_xformed - model_transform[:, None, None, :3, 3]) @ model_transform[:, None, :3, :3]


@torch.no_grad()
def xyz_to_nn_target(xyz: Tensor, noc_scale_xfd: Tensor = None, *, noc_mask=None, model_transform=None, bit_cnt=None):
    """Inverse of `nn_out_to_xyz` used for label preparation (`losses.py:49-67`): object coordinates (B,H,W,3) in mm -> the
    network target, either normalised coordinates (B,3,H,W) or (binary code planes, raw bits) of the ZebraPose heads."""
    xformed = xyz
    if model_transform is not None:
        xformed = xyz @ model_transform[:, None, :3, :3].mT + model_transform[:, None, None, :3, 3]
        if noc_mask is not None:
            xformed = xformed * noc_mask.unsqueeze(-1)
    noc = xformed / noc_scale_xfd[:, None, None, :]
    if bit_cnt is None:
        assert model_transform is None, "coordinate transform not implemented for continuous xyz output"
        return noc.permute(0, 3, 1, 2), None
    return floatbits.nn_noc2target(noc, bit_cnt)


def dense_pnp_matching_from_noc_bin(noc_bin_out_logits: Tensor, noc_bin_gt_raw: Tensor, weights_out: Tensor, valid_msk_full: Tensor,
                                    noc_mask: Tensor, noc_scale: Tensor, gt_dict: dict, sample=2, top_left=None):
    """`losses.py:163-184`: strided sub-sampling of a binary-code head; the 3D points are decoded on the sampled pixels only."""
    top, left = np.random.randint(0, sample, size=2) if top_left is None else top_left
    uv_grid = xforms.gen_uv(weights_out.shape[-2:], weights_out.device)
    pts2d = uv_grid[..., top::sample, left::sample, :].flatten(start_dim=-3, end_dim=-2)
    inv_std2d = weights_out[..., top::sample, left::sample].flatten(start_dim=-2).mT
    valid_msk = valid_msk_full[..., top::sample, left::sample].flatten(start_dim=-2)
    pts3d = _decode_bin_points(noc_bin_out_logits, noc_bin_gt_raw, noc_mask, noc_scale, gt_dict, sample, (top, left))
    return pts2d.expand_as(inv_std2d), inv_std2d, pts3d, valid_msk


def _decode_bin_points(logits, raw_bits, noc_mask, noc_scale, gt_dict, sample, top_left):
    T = gt_dict.get("model_transform", None)
    if logits.is_cuda and logits.dtype in _lib.MAP_DTYPES and noc_scale.dtype == torch.float32 and (T is None or T.dtype == torch.float32):  # fp32 / fp16 / bf16 logits, read natively
        # decode, `noc * noc_scale` and the model transform `(xyz - T[:, :3, 3]) @ T[:, :3, :3]` in ONE launch each way
        return floatbits.decode_with_gt_strided(logits, raw_bits, gt_dict["bit_cnt"], noc_mask, sample=sample, top_left=top_left,
                                                out_scale=noc_scale, out_xform=T)
    noc = floatbits.decode_with_gt_strided(logits, raw_bits, gt_dict["bit_cnt"], noc_mask, sample=sample, top_left=top_left)  # (B,N,3)
    xyz = noc * noc_scale.unsqueeze(-2)
    if T is not None:
        xyz = (xyz - T[:, None, :3, 3]) @ T[:, :3, :3]
    return xyz


class Loss_xyz_bin(nn.Module):
    """`losses.py:196-216`: per-bit weighted BCE on the code logits with an EMA histogram of per-bit Hamming errors
    (`histogram` buffer: rides in the checkpoint as `loss_fn.xyz_bin_loss_fn.histogram`)."""

    def __init__(self, total_bit_cnt: int, momentum=0.05, group=None) -> None:
        super().__init__()
        self.register_buffer("histogram", torch.full((total_bit_cnt,), 0.5))
        self.momentum = momentum
        self.group = group  # the batch is sharded over this process group: the histogram is the WHOLE batch's (collective 4)

    def _sharded(self) -> bool:
        import torch.distributed as dist

        # a group of one rank takes the sharded form too (its all-reduce is the identity): the same launches whatever the GPU count
        return self.group is not None and dist.is_available() and dist.is_initialized()

    def forward(self, noc_xyz_bin_logits: Tensor, noc_xyz_bin_gt: Tensor, msk_vis_logits: Tensor):
        sharded = self._sharded()
        if dense_aux.fused_path_ok(noc_xyz_bin_logits, msk_vis_logits, self.histogram) and self.histogram.numel() <= 128 and not _TORCH_SHARDED_ROUTE:
            # one pass over the logits instead of ~12 (lc_amd/csrc/lc_dense_aux.hip); the histogram buffer is updated in place.  Sharded: the same
            # pass, split around the all-reduce of the C + 1 counts -- the same kernels and numerics for every map type and GPU count
            return dense_aux.xyz_bin_loss(noc_xyz_bin_logits, noc_xyz_bin_gt, msk_vis_logits, self.histogram, self.momentum,
                                          group=self.group if sharded else None)
        # the reference's torch formulas (maps that are not HIP tensors: the world-size-2 gloo tests on the CPU)
        msk_hard = msk_vis_logits > 0  # the comparisons need no up-cast of a 16-bit map
        hamm = (noc_xyz_bin_logits > 0).logical_xor(noc_xyz_bin_gt.to(torch.bool)).logical_and(msk_hard)
        counts = torch.cat((hamm.sum([0, 2, 3]), msk_hard.sum().reshape(1)))  # int64, like the reference's sums
        if sharded:
            # The reference is one process: its histogram update (losses.py:203-208) sees the Hamming errors and the visible pixels of the whole
            # batch.  Per-rank updates would let the checkpointed buffer -- and with it the bit weights and the gradient -- depend on the GPU
            # count, so the C error counts and the pixel count are all-reduced as integers (exact at any batch size) and every rank applies the
            # single process' update.  (The loss itself stays the rank's mean: DDP averages the ranks.)
            import torch.distributed as dist

            dist.all_reduce(counts, group=self.group)
        hist = counts[:-1] / (counts[-1] + 1)
        self.histogram.mul_(1 - self.momentum).add_(hist * self.momentum)
        hist_soft = torch.minimum(self.histogram, 0.51 - self.histogram)
        bin_weights = (hist_soft * 3).softmax(dim=-1)
        masked = noc_xyz_bin_logits * msk_hard
        if masked.dtype in (torch.float16, torch.bfloat16):  # 16-bit heads: the BCE in fp32, as the fused kernel computes it
            masked = masked.float()
        loss_raw = F.binary_cross_entropy_with_logits(masked, noc_xyz_bin_gt.float(), reduction="none")  # losses.py:213: the targets in float32 whatever the logits' type
        return (loss_raw.mean([0, 2, 3]) * bin_weights).sum(-1).mean()


def dense_pnp_matching_from_xyz(xyz_out: Tensor, weights_out: Tensor, valid_msk_full: Tensor, xyz_scale: Tensor, sample=2,
                                top_left=None):
    """Strided sub-sampling of the dense head into 2D-3D correspondences (`losses.py:142-161`).

    xyz_out (*,3,H,W), weights_out (*,2,H,W), valid_msk_full (*,H,W)|None, xyz_scale (*,3)|None
    -> pts2d (*,N,2) (pixel grid, constant), inv_std2d (*,N,2), pts3d (*,N,3), valid (*,N)|None, N = ceil(H/s)*ceil(W/s).
    The random phase uses `np.random.randint` like the reference so seeded runs line up.
    """
    top, left = np.random.randint(0, sample, size=2) if top_left is None else top_left
    uv_grid = xforms.gen_uv(xyz_out.shape[-2:], xyz_out.device)
    pts2d = uv_grid[..., top::sample, left::sample, :].flatten(start_dim=-3, end_dim=-2)
    inv_std2d = weights_out[..., top::sample, left::sample].flatten(start_dim=-2).mT
    pts3d = xyz_out[..., top::sample, left::sample].flatten(start_dim=-2).mT
    if xyz_scale is not None:
        pts3d = pts3d * xyz_scale.unsqueeze(-2)
    valid_msk = valid_msk_full[..., top::sample, left::sample].flatten(start_dim=-2) if valid_msk_full is not None else None
    return pts2d.expand_as(inv_std2d), inv_std2d, pts3d, valid_msk


class Loss_seg_L1(nn.Module):
    """`losses.py:219-236`."""

    def forward(self, input: Tensor, target: Tensor, weight: Tensor = None, reduction: str = "mean"):
        err = (input.sigmoid() - target).abs()
        if weight is not None:
            err = err * weight
        if reduction != "mean":
            raise NotImplementedError
        return err.mean()


def pose_loss_factor(cfg, step, steps_per_epoch) -> float:
    """Warm-up ramp of the LC loss (`losses.py:270-274,296-302`)."""
    full_step = max(cfg.get("pose_loss_start_step", 0), cfg.get("pose_loss_start_epoch", 0) * steps_per_epoch)
    nz_step = cfg.get("loss_pose_nz_step", 0)
    return max(0, min((step - nz_step + 1) / (max(full_step - nz_step, 0) + 1e-5), 1))


class Loss_fn(nn.Module):
    """Drop-in for `losses.Loss_fn` (`losses.py:239-386`): `forward(gt_dict, out_dict, epoch, step, steps_per_epoch)`
    -> `(loss_dict, w_loss_dict)`.  `group`: optional process group; when the batch is sharded over ranks the
    NormClippers all-reduce their squared norm over it and `Loss_xyz_bin` the error counts of its histogram (collectives 2 and 4); `shard_loss_scale`: lc_amd/grad.py (1 / world_size under
    DistributedDataParallel)."""

    def __init__(self, cfg, cfg_global, total_bit_cnt=0, group=None, shard_loss_scale=1.0) -> None:
        super().__init__()
        self.cfg = cfg
        pose_cfg = cfg.pose_loss_cfg
        kw = dict(group=group, shard_loss_scale=shard_loss_scale)
        self.weight_grad_clipper = NormClipper(**kw) if pose_cfg.get("clip_weight_grad", True) else None
        self.scale_grad_clipper = NormClipper(rel_thresh=2, **kw) if pose_cfg.get("clip_scale_grad", False) else None
        self.pts_grad_clipper = NormClipper(rel_thresh=2, **kw) if pose_cfg.get("clip_pts_grad", False) else None
        self.cfg_global = cfg_global
        if total_bit_cnt > 0:
            self.xyz_bin_loss_fn = Loss_xyz_bin(total_bit_cnt, group=group)
        seg_loss_type = cfg.get("seg_loss_type", "BCE")
        self.seg_loss_type = seg_loss_type.lower()
        if seg_loss_type.lower() == "bce":
            self.seg_loss_fn = F.binary_cross_entropy_with_logits
        elif seg_loss_type.lower() == "l1":
            self.seg_loss_fn = Loss_seg_L1()

    def forward(self, gt_dict, out_dict, epoch, step, steps_per_epoch):
        cfg = self.cfg
        msk_noc, msk_vis = itemgetter("msk_noc", "msk_vis")(gt_dict)
        loss_dict = {}

        if "pts2d" in out_dict:  # sparse (keypoint) case, losses.py:267-279
            loss_kpts = self.sparse_kpt_loss(cfg, gt_dict, out_dict)
            loss_dict["loss_kpts"] = loss_kpts
            if cfg.get("w_loss_pose", 0) > 0:
                factor = pose_loss_factor(cfg, step, steps_per_epoch)
                loss_pose = self.sparse_pose_loss(cfg, gt_dict, out_dict)
                loss_dict["loss_pose"] = factor * loss_pose + (1 - factor) * loss_kpts
            w_loss_dict = {k: v * cfg.get("w_" + k, 0) for k, v in loss_dict.items() if cfg.get("w_" + k, 0) > 0}
            return loss_dict, w_loss_dict

        # dense case, losses.py:281-316
        if "xyz_noc_bin" in out_dict:  # zebra-pose structure (losses.py:289-291)
            loss_dict["loss_noc_bin"] = self.xyz_bin_loss_fn(out_dict["xyz_noc_bin"], gt_dict["xyz_noc_bin_tgt"],
                                                             out_dict["msk_vis_logits"])
        factor = pose_loss_factor(cfg, step, steps_per_epoch)
        xyz_noc = out_dict.get("xyz_noc")
        weight_logits = out_dict["xyz_weight_logits"] if factor != 1 else None
        if dense_aux.fused_path_ok(xyz_noc, out_dict["msk_vis_logits"], weight_logits, msk_vis, gt_dict.get("xyz_noc_tgt") if xyz_noc is not None else None):
            # loss_noc, loss_seg and (during the warm-up) loss_weight_seg: one launch each way (lc_amd/dense_aux.py)
            loss_noc, loss_seg, loss_weight_seg = dense_aux.dense_aux_losses(
                xyz_noc, msk_noc if xyz_noc is not None else None, gt_dict["xyz_noc_tgt"] if xyz_noc is not None else None,
                out_dict["msk_vis_logits"], msk_vis, weight_logits, self.seg_loss_type)
            if xyz_noc is not None:
                loss_dict["loss_noc"] = loss_noc
            loss_dict["loss_seg"] = loss_seg
        else:  # half-precision heads: the reference's torch formulas
            if xyz_noc is not None:
                noc_msked, noc_gt = xyz_noc * msk_noc.unsqueeze(-3), gt_dict["xyz_noc_tgt"]
                loss_dict["loss_noc"] = F.l1_loss(noc_msked, noc_gt, reduction="mean")
            loss_dict["loss_seg"] = self.seg_loss_fn(out_dict["msk_vis_logits"], msk_vis.unsqueeze(-3), reduction="mean")
            if factor != 1:
                msk_vis_tgt = msk_vis.unsqueeze(-3).expand_as(weight_logits)
                loss_weight_seg = self.seg_loss_fn(weight_logits, msk_vis_tgt, reduction="mean")

        loss_pose = self.dense_pose_loss(cfg.pose_loss_cfg, gt_dict, out_dict)
        if factor != 1:
            loss_pose = factor * loss_pose + (1 - factor) * loss_weight_seg
        loss_dict["loss_pose"] = loss_pose

        loss_dict = {k: v.mean() if len(v.shape) > 0 else v for k, v in loss_dict.items() if isinstance(v, Tensor)}
        w_loss_dict = {k: v * cfg.get("w_" + k, 0) for k, v in loss_dict.items() if cfg.get("w_" + k, 0) > 0}
        return loss_dict, w_loss_dict

    def sparse_kpt_loss(self, cfg, gt_dict, out_dict):
        """Laplace NLL of the keypoints, mean(log sigma + |u - proj|/sigma) (`losses.py:318-326`): one fused launch."""
        pts2d, pts2d_std = itemgetter("pts2d", "pts2d_std")(out_dict)
        pose_best, K, pts3d = itemgetter("pose_best", "out_K", "pts3d")(gt_dict)
        return kpt_nll_mean(K, pose_best, pts3d, pts2d, pts2d_std)

    def sparse_pose_loss(self, cfg, gt_dict, out_dict):
        """`losses.py:329-334`."""
        pts2d, pts2d_std = itemgetter("pts2d", "pts2d_std")(out_dict)
        pose_best, K, pts3d, bbox_3d = itemgetter("pose_best", "out_K", "pts3d", "bbox_3d")(gt_dict)
        return Loss_cov_mixed(K, pose_best, pts3d, pts2d, 1 / pts2d_std, None, bbox_3d=bbox_3d).mean()

    def dense_pose_loss(self, cfg, gt_dict, out_dict):
        """`losses.py:336-386` (GDR-Net continuous-xyz structure or ZebraPose binary-code structure)."""
        noc_scale = gt_dict["noc_scale"]
        pose_best, K, bbox_3d = itemgetter("pose_best", "out_K", "bbox_3d")(gt_dict)

        xyz_weight_logits: Tensor = out_dict["xyz_weight_logits"]
        if self.weight_grad_clipper is not None and xyz_weight_logits.requires_grad:
            xyz_weight_logits.register_hook(lambda grad: self.weight_grad_clipper.clip(grad))
        xyz_weights_scale: Tensor = out_dict["xyz_weights_scale"]
        if self.scale_grad_clipper is not None and xyz_weights_scale.requires_grad:
            xyz_weights_scale.register_hook(lambda grad: self.scale_grad_clipper.clip(grad))

        # joint softmax over all 2*H*W logits x per-sample scale (losses.py:355-356) + strided sub-sampling with random
        # phase (losses.py:142-161): one fused HIP launch each way (lc_amd/dense.py)
        sample = cfg.get("dense_sample", 2)
        assert ("xyz_noc" in out_dict) != ("xyz_noc_bin" in out_dict)  # either structure, not both (losses.py:360)
        # sub-sampling phase: drawn from np.random like the reference (losses.py:152,169) unless a caller that replays this step
        # as a hipGraph pinned it (lc_amd/graphs.py draws it outside the graph and keeps one graph per phase)
        phase = getattr(self, "_forced_phase", None)
        if "xyz_noc" in out_dict:
            den_pts2d, den_inv_std2d, den_pts3d = dense_front_end(out_dict["xyz_noc"], xyz_weight_logits, xyz_weights_scale, noc_scale,
                                                                  sample=sample, top_left=phase)
        else:
            top_left = tuple(int(v) for v in np.random.randint(0, sample, size=2)) if phase is None else phase  # losses.py:169
            den_pts2d, den_inv_std2d, _ = dense_front_end(None, xyz_weight_logits, xyz_weights_scale, None, sample=sample, top_left=top_left)
            den_pts3d = _decode_bin_points(out_dict["xyz_noc_bin"], gt_dict["xyz_noc_bin_raw"], gt_dict["msk_noc"], noc_scale, gt_dict,
                                           sample, top_left)
        den_valid_msk = None  # losses.py:370 passes a mask of ones: every correspondence valid, which is what no mask means to the kernel
        if self.pts_grad_clipper is not None and den_pts3d.requires_grad:
            den_pts3d.register_hook(lambda grad: self.pts_grad_clipper.clip(grad))

        loss_cov = Loss_cov_mixed(K, pose_best, den_pts3d, den_pts2d, den_inv_std2d, den_valid_msk, bbox_3d=bbox_3d,
                                  max_err_len=cfg.get("max_err_len", 32))
        return loss_cov.mean()
