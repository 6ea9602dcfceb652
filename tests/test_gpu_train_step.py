"""The example training step (backbone -> fused head -> Loss_fn -> backward -> optimiser) runs and learns on one GPU."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_train_step_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_sparse_ddp.py"), "--steps", "6", "--batch", "4",
                          "--sparse-cnt", "16", "--width", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("step")]
    assert len(lines) == 6 and "median step" in out.stdout
    assert all("nan" not in l.lower() for l in lines)


def test_graphed_sparse_loss_step_equals_eager_on_new_inputs():
    """Forward + backward of the sparse Loss_fn step replayed as hipGraphs: same loss values and gradients as the eager call."""
    import warnings

    from lc_amd.config import AttrDict
    from lc_amd.graphs import graphed_sparse_loss
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import SPARSE_CFG, sparse_inputs

    dev = torch.device("cuda:0")
    fn = Loss_fn(AttrDict(SPARSE_CFG), AttrDict(), 0).to(dev)

    def inputs(seed):
        gt, out = sparse_inputs(B=32, N=16, seed=seed)
        return {k: v.to(dev) for k, v in gt.items()}, {k: v.to(dev) for k, v in out.items()}

    gt0, out0 = inputs(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        graphed = graphed_sparse_loss(fn, gt0, out0, 1, 1000, 10)
        for seed in (0, 1, 2):
            gt, out = inputs(seed)
            u, s = out["pts2d"].clone().requires_grad_(True), out["pts2d_std"].clone().requires_grad_(True)
            total, lk, lp = graphed(u, s, gt["out_K"], gt["pose_best"], gt["pts3d"], gt["bbox_3d"])
            gu, gs = torch.autograd.grad(total, (u, s))
            u2, s2 = out["pts2d"].clone().requires_grad_(True), out["pts2d_std"].clone().requires_grad_(True)
            ld, wd = fn(gt, dict(pts2d=u2, pts2d_std=s2), 1, 1000, 10)
            ru, rs = torch.autograd.grad(sum(wd.values()), (u2, s2))
            assert torch.equal(total, sum(wd.values())) and torch.equal(lk, ld["loss_kpts"]) and torch.equal(lp, ld["loss_pose"])
            assert torch.equal(gu, ru) and torch.equal(gs, rs)
