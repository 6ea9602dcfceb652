"""The example training step (backbone -> fused head -> Loss_fn -> backward -> optimiser) runs and learns on one GPU."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--graphs"]], ids=["eager", "graphs"])
def test_example_train_step_runs(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_sparse_ddp.py"), "--steps", "8", "--batch", "4",
                          "--sparse-cnt", "16", "--width", "16"] + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("step")]
    assert len(lines) == 8 and "median step" in out.stdout
    assert all("nan" not in l.lower() for l in lines)


@pytest.mark.parametrize("extra", [[], ["--bin"]], ids=["xyz", "binary-code"])
def test_example_dense_train_step_runs(extra):
    """BASELINE configs[4] plumbing: dense heads (continuous xyz / ZebraPose codes), fp16 autocast backbone, GradScaler, clippers."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_dense_ddp.py"), "--steps", "6", "--batch", "4", "--width", "16"]
                         + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "median step" in out.stdout


@pytest.mark.parametrize("kind", ["sparse", "dense", "bin"])
def test_graphed_loss_step_equals_eager_on_new_inputs(kind):
    """Forward + backward of a Loss_fn step replayed as hipGraphs: same loss values, gradients and NormClipper trajectory as
    the eager calls, step after step (dense: one graph per sub-sampling phase, drawn from np.random like the reference)."""
    import warnings

    import numpy as np

    from lc_amd.config import AttrDict
    from lc_amd.graphs import GraphedLoss
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import BIN_CFG, DENSE_CFG, SPARSE_CFG, bin_inputs, dense_inputs, sparse_inputs

    dev = torch.device("cuda:0")
    cfg = {"sparse": SPARSE_CFG, "dense": DENSE_CFG, "bin": BIN_CFG}[kind]
    make = {"sparse": lambda s: sparse_inputs(B=32, N=16, seed=s), "dense": lambda s: dense_inputs(B=4, H=16, W=16, seed=s),
            "bin": lambda s: bin_inputs(B=4, H=16, W=16, seed=s)}[kind]
    bits = 17 if kind == "bin" else 0

    def inputs(seed):
        gt, out = make(seed)
        return ({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}, {k: v.to(dev) for k, v in out.items()})

    eager_fn = Loss_fn(AttrDict(cfg), AttrDict(), bits).to(dev)
    graph_fn = Loss_fn(AttrDict(cfg), AttrDict(), bits).to(dev)
    gt0, out0 = inputs(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        graphed = GraphedLoss(graph_fn, gt0, out0, 1, 1000, 10)
        for i, seed in enumerate((0, 1, 2, 3, 4, 5)):
            gt, out = inputs(seed)
            la = {k: v.clone().requires_grad_(True) for k, v in out.items()}
            lb = {k: v.clone().requires_grad_(True) for k, v in out.items()}
            np.random.seed(100 + i)
            ld, wd = graphed(gt, la)
            ga = torch.autograd.grad(sum(wd.values()), list(la.values()), allow_unused=True)
            np.random.seed(100 + i)
            rd, rw = eager_fn(gt, lb, 1, 1000, 10)
            gb = torch.autograd.grad(sum(rw.values()), list(lb.values()), allow_unused=True)
            assert list(ld) == list(rd) and list(wd) == list(rw)
            assert all(torch.equal(ld[k], rd[k]) for k in rd) and all(torch.equal(wd[k], rw[k]) for k in rw)
            for x, y in zip(ga, gb):
                assert (x is None) == (y is None) and (x is None or torch.equal(x, y))
            for (ka, va), (kb, vb) in zip(graph_fn.state_dict().items(), eager_fn.state_dict().items()):
                assert ka == kb and torch.equal(va, vb), (i, ka, va, vb)
