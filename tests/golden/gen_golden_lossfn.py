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


from lc_amd.synth import bin_inputs, dense_inputs, sparse_inputs  # noqa: E402,F401  (the input builders live in the package: the benches use them too)


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
