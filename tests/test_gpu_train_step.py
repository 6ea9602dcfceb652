"""The example training step (backbone -> fused head -> Loss_fn -> backward -> optimiser) runs and learns on one GPU."""
import os
import subprocess
import sys

import numpy as np
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


@pytest.mark.parametrize("extra", [[], ["--bin"], ["--dtype", "bf16"], ["--dtype", "bf16", "--bin"], ["--dtype", "bf16", "--graphs"],
                                   ["--dtype", "bf16", "--width", "64", "--batch", "8", "--steps", "5"],
                                   ["--zlmo"], ["--zlmo", "--graphs", "--steps", "14"], ["--zlmo", "--width", "64", "--batch", "8", "--steps", "5"]],
                         ids=["xyz-fp16", "binary-code-fp16", "xyz-bf16", "binary-code-bf16", "xyz-bf16-graphed-loss", "xyz-bf16-resnet34-width",
                              "binary-code-fp16-zlmo-shape", "binary-code-fp16-zlmo-shape-graphed-loss", "binary-code-fp16-zlmo-shape-resnet34-width"])
def test_example_dense_train_step_runs(extra):
    """BASELINE configs[2]/[4]: dense heads (continuous xyz / ZebraPose codes), fp16 (GradScaler) or bf16 autocast backbone,
    clippers; Loss_fn eager or replayed as hipGraphs; `resnet34-width` runs the 64-128-256-512-channel backbone in bf16.
    `zlmo-shape`: configs/zlmo.yaml's own step -- output-stride-8 dilated trunk + atrous pyramid (examples/os8_trunk.py), 128x128 maps, 7+7+7
    code planes, dense_sample 3 => N = 1849, fp16 + GradScaler, zlmo.yaml:74-83's loss block; graphed: one hipGraph per sub-sampling phase (nine)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_dense_ddp.py"), "--steps", "6", "--batch", "4", "--width", "16"]
                         + extra, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "median step" in out.stdout and "nan" not in out.stdout.lower()
    if "--zlmo" in extra:
        assert "os8 trunk, 128x128 maps, binary-code (21 planes) dense head, N=1849 correspondences per sample" in out.stdout


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


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["dense", "bin"])
def test_loss_fn_dense_branch_takes_half_precision_heads(kind, dtype):
    """A mixed-precision backbone hands fp16 / bf16 head outputs straight to Loss_fn (BASELINE configs 3 and 5): the NormClipper
    hooks sit on those half tensors, so they must return gradients in the same dtype (autograd rejects a hook that changes
    it), and the step must agree with the fp32 step on the same (rounded) values to half-precision accuracy."""
    from lc_amd.config import AttrDict
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import BIN_CFG, DENSE_CFG, bin_inputs, dense_inputs

    dev = torch.device("cuda:0")
    cfg = dict({"dense": DENSE_CFG, "bin": BIN_CFG}[kind])
    cfg["pose_loss_cfg"] = dict(cfg["pose_loss_cfg"], clip_weight_grad=True, clip_scale_grad=True, clip_pts_grad=True)
    gt, out = (dense_inputs if kind == "dense" else bin_inputs)(B=4, H=16, W=16, seed=3)
    gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
    bits = 17 if kind == "bin" else 0
    res = {}
    for name, dt in (("half", dtype), ("ref", torch.float32)):
        fn = Loss_fn(AttrDict(cfg), AttrDict(), bits).to(dev)
        o = {k: v.to(dev).to(dtype).to(dt).requires_grad_(True) for k, v in out.items()}  # same rounded values in both runs
        np.random.seed(5)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            ld, wd = fn(gt, o, 1, 1000, 10)
            total = sum(wd.values())
            total.backward()
            torch.cuda.synchronize()
        if name == "half":  # the heads' maps are read in their own type: not one dtype-cast launch in the step, forward or backward
            casts = [e for e in prof.events() if e.name == "aten::_to_copy"]
            assert not casts, [(e.name, e.input_shapes) for e in casts]
        assert all(v.grad is None or v.grad.dtype == dt for v in o.values())
        res[name] = (float(total), {k: v.grad.float() for k, v in o.items() if v.grad is not None}, float(fn.weight_grad_clipper.max_norm))
    assert abs(res["half"][0] - res["ref"][0]) <= 2e-2 * max(1.0, abs(res["ref"][0]))
    assert res["half"][1].keys() == res["ref"][1].keys()
    for k, g in res["ref"][1].items():
        assert torch.isfinite(res["half"][1][k]).all()
        err = (res["half"][1][k] - g).abs().max() / g.abs().max().clamp_min(1e-20)
        assert err <= (5e-2 if dtype == torch.bfloat16 else 1e-2), (k, float(err))
    assert abs(res["half"][2] - res["ref"][2]) <= 5e-2 * abs(res["ref"][2])


def test_graphed_loss_never_replays_a_stale_warmup_blend():
    """GraphedLoss bakes the warm-up factor (losses.py:272-276) into its graphs: called with `step` inside the ramp it must run
    the step itself (new blend every step), and at the plateau it must capture for THAT factor -- always equal to eager."""
    import warnings

    from lc_amd.config import AttrDict
    from lc_amd.graphs import GraphedLoss
    from lc_amd.losses import Loss_fn, pose_loss_factor
    from tests.golden.gen_golden_lossfn import SPARSE_CFG, sparse_inputs

    dev = torch.device("cuda:0")
    cfg = dict(SPARSE_CFG, pose_loss_start_step=40, loss_pose_nz_step=10, w_loss_pose=0.7)
    eager_fn = Loss_fn(AttrDict(cfg), AttrDict()).to(dev)
    graph_fn = Loss_fn(AttrDict(cfg), AttrDict()).to(dev)
    gt, out = sparse_inputs(B=16, N=16, seed=0)
    gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
    out = {k: v.to(dev) for k, v in out.items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        graphed = GraphedLoss(graph_fn, gt, out, 0, 0, 10)  # captured BEFORE the ramp (factor 0)
        seen = set()
        for step in (0, 5, 12, 25, 39, 40, 41, 100):
            f = pose_loss_factor(AttrDict(cfg), step, 10)
            seen.add(0 if f == 0 else 1 if f == 1 else 2)
            la = {k: v.clone().requires_grad_(True) for k, v in out.items()}
            lb = {k: v.clone().requires_grad_(True) for k, v in out.items()}
            ld, wd = graphed(gt, la, step=step)
            rd, rw = eager_fn(gt, lb, 0, step, 10)
            assert all(torch.equal(ld[k], rd[k]) for k in rd), (step, f, {k: (float(ld[k]), float(rd[k])) for k in rd})
            ga = torch.autograd.grad(sum(wd.values()), list(la.values()))
            gb = torch.autograd.grad(sum(rw.values()), list(lb.values()))
            assert all(torch.equal(x, y) for x, y in zip(ga, gb)), step
    assert seen == {0, 1, 2}  # the sweep did cross both plateaus and the ramp
