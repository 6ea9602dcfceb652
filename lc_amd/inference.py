"""Test-time pose recovery of `test.py:39-136` (`solve_pnp`, `solve_pnp_dense`, `quantile_msk`) with every stage on the GPU:
network output -> (decode) -> weights -> point selection -> RANSAC-P3P initialiser (`lc_amd.pnp.gpu_solver`, in place of
cv2.solvePnPRansac) -> weighted LM (`lc_amd.pnp.cer_solver`).  Returns the same dict the reference returns
({'weighted': states, 'ransac': states, 'weighted-filtered': states}, in the same key order)."""
from __future__ import annotations

from operator import itemgetter

import numpy as np
import torch
from torch import Tensor

from .dense import dense_front_end
from .losses import nn_out_to_xyz
from .pnp import cer_solver, gpu_solver


def quantile_msk(den_inv_std2d: Tensor, quantile):
    """`test.py:39-45`: keep the points whose summed weight is at or above the per-sample quantile."""
    weights = den_inv_std2d.sum(dim=-1)
    q = torch.quantile(weights, quantile, dim=1, keepdim=True)
    if isinstance(quantile, Tensor):
        q = torch.diagonal(q, dim1=0, dim2=1).mT
    return weights >= q


@torch.no_grad()
def solve_pnp(cfg, out_dict, gt_dict):
    """`test.py:47-64`."""
    if "pts2d" not in out_dict:
        return solve_pnp_dense(cfg, out_dict, gt_dict)
    K, pts3d = itemgetter("out_K", "pts3d")(gt_dict)
    pts2d, pts2d_std = itemgetter("pts2d", "pts2d_std")(out_dict)
    inv_cov2d = 1 / (pts2d_std ** 2)
    reproj = 2
    if cfg.get("rel_reproj_err", False):
        reproj = 2 / gt_dict["out_pix_scale"]
    invalids, cv_states, inliers = gpu_solver.solve(K, pts3d, pts2d, reprojectionError=reproj)
    cv_states = torch.stack([st.to(torch.float32) for st in cv_states])
    weighted = cer_solver.solve(K, pts3d, pts2d, inv_cov2d, cv_states, num_workers=4, filter_input_nan=True)[1]
    return dict([("weighted", weighted), ("ransac", cv_states)])


@torch.no_grad()
def solve_pnp_dense(cfg, out_dict, gt_dict):
    """`test.py:67-136`."""
    K = gt_dict["out_K"]
    seg_msk = torch.sigmoid(out_dict["msk_vis_logits"]) > cfg.get("seg_thresh", 0.5)
    sample = cfg.get("dense_sample", 2)
    nn_out = out_dict["xyz_noc"] if "xyz_noc" in out_dict else out_dict["xyz_noc_bin"]
    xyz_out = nn_out_to_xyz(nn_out, gt_dict["noc_scale"], model_transform=gt_dict.get("model_transform", None),
                            bit_cnt=gt_dict.get("bit_cnt", None), inference=True)  # (B,H,W,3)
    # joint softmax x scale + strided sub-sampling at phase (0,0) (test.py:85-92): the fused front end
    den_pts2d, den_inv_std2d, den_pts3d = dense_front_end(xyz_out.permute(0, 3, 1, 2), out_dict["xyz_weight_logits"],
                                                          out_dict["xyz_weights_scale"], None, sample=sample, top_left=(0, 0))
    seg_valid_mask = seg_msk.squeeze(-3)[..., 0::sample, 0::sample].flatten(start_dim=-2)
    den_inv_cov2d = den_inv_std2d ** 2

    sel = cfg.dense_point_select
    if sel == "mask":
        den_valid_msk = seg_valid_mask
    elif sel == "quantile":
        den_valid_msk = quantile_msk(den_inv_std2d, cfg.quantile)
    elif sel == "quantile_in_mask":
        vis_ratio = seg_valid_mask.float().mean(dim=-1)
        quantile = 1 - (1 - cfg.quantile) * vis_ratio
        den_valid_msk = quantile_msk(den_inv_std2d * seg_valid_mask[..., None], quantile) * seg_valid_mask
    else:
        raise ValueError(f"unknown dense_point_select {sel!r}")
    valid_index_lst = [v.nonzero()[:, 0] for v in den_valid_msk]

    def min_len_index(idx: Tensor, src_len, min_len):
        n, dev, dtype = min_len - len(idx), idx.device, idx.dtype
        return idx if n <= 0 else torch.cat((idx, torch.from_numpy(np.random.choice(src_len, n, n > src_len)).to(dev, dtype)))

    def select_valid(src_tensors, idx, min_cnt=4):
        return [t[min_len_index(i, len(t), min_cnt)] if len(t) > min_cnt else t for t, i in zip(src_tensors, idx)]

    reproj = 3
    if cfg.get("rel_reproj_err", False):
        reproj = 2 / gt_dict["out_pix_scale"]
    pts3d, pts2d = (select_valid(t, valid_index_lst) for t in (den_pts3d, den_pts2d))
    invalids, cv_states, inliers = gpu_solver.solve(K, pts3d, pts2d, reprojectionError=reproj)
    cv_states = torch.stack([st.to(torch.float32) for st in cv_states])

    res = []
    solvers = cfg.solvers
    if "weighted" in solvers:
        inv_cov2d = select_valid(den_inv_cov2d, valid_index_lst)
        res.append(("weighted", cer_solver.solve(K, pts3d, pts2d, inv_cov2d, cv_states, num_workers=4, filter_input_nan=True)[1]))
    if "weighted_filtered" in solvers:
        filtered_valid_idx = select_valid(valid_index_lst, inliers)
        p3, p2, ic = (select_valid(t, filtered_valid_idx) for t in (den_pts3d, den_pts2d, den_inv_cov2d))
        res.append(("weighted-filtered", cer_solver.solve(K, p3, p2, ic, cv_states, num_workers=4, filter_input_nan=True)[1]))
    if "ransac" in solvers:
        res.append(("ransac", cv_states))
    return dict(res[::-1])
