"""hipGraph replay of the launch-bound pieces around the kernels (MI355X guideline: capture launch-bound inner loops).

The sparse heads' `Loss_fn` step is ~25 short launches forward + backward for < 20 us of kernel time, i.e. bound by the host's
launch rate; captured with `torch.cuda.make_graphed_callables` it replays as two graphs (forward, backward):
B=256, N=64 on one MI355X: 338 us eager -> 103 us graphed, loss and gradients bit-identical (`scripts/ubench/graph_lossfn.py`).
The test-time counterpart is `lc_amd.inference.GraphedSolvePnP`.
"""
from __future__ import annotations

import torch

from .inference import GraphedSolvePnP  # noqa: F401  (re-exported)


def graphed_sparse_loss(loss_fn, gt_dict: dict, out_dict: dict, epoch: int, step: int, steps_per_epoch: int):
    """Capture `loss_fn(gt_dict, out_dict, epoch, step, steps_per_epoch)` of the sparse branch for the shapes of the example dicts.

    Returns `f(pts2d, pts2d_std, out_K, pose_best, pts3d, bbox_3d) -> (total, loss_kpts, loss_pose)`; `total` is the weighted sum
    the training loop back-propagates (`sum(w_loss_dict.values())`), differentiable w.r.t. `pts2d` and `pts2d_std`.
    The warm-up blending factor of `losses.py:272-276` is a Python float and is frozen at its value for `step`: capture after
    the ramp (or capture again when it changes).
    """
    if "pts2d" not in out_dict:
        raise ValueError("graphed_sparse_loss: the sparse branch needs out_dict['pts2d'] / ['pts2d_std']")
    rest = {k: v for k, v in gt_dict.items() if k not in ("out_K", "pose_best", "pts3d", "bbox_3d")}

    def run(pts2d, pts2d_std, out_K, pose_best, pts3d, bbox_3d):
        gt = dict(rest, out_K=out_K, pose_best=pose_best, pts3d=pts3d, bbox_3d=bbox_3d)
        loss_dict, w_loss_dict = loss_fn(gt, dict(pts2d=pts2d, pts2d_std=pts2d_std), epoch, step, steps_per_epoch)
        zero = pts2d.new_zeros(())
        return sum(w_loss_dict.values()), loss_dict.get("loss_kpts", zero), loss_dict.get("loss_pose", zero)

    sample = (out_dict["pts2d"].detach().clone().requires_grad_(True), out_dict["pts2d_std"].detach().clone().requires_grad_(True),
              gt_dict["out_K"].detach().clone(), gt_dict["pose_best"].detach().clone(), gt_dict["pts3d"].detach().clone(),
              gt_dict["bbox_3d"].detach().clone())
    return torch.cuda.make_graphed_callables(run, sample)
