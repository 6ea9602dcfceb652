"""GPU parity of the fused dense front end against the reference's op sequence (oracle/dense_oracle.py)."""
import re

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,sample,tl", [(3, 16, 16, 2, (0, 0)), (2, 64, 64, 2, (1, 1)), (2, 64, 64, 2, (1, 0)), (2, 30, 50, 3, (2, 1)),
                                              (1, 128, 128, 3, (0, 2)), (4, 9, 7, 1, (0, 0))])
def test_dense_front_end_vs_oracle(B, H, W, sample, tl):
    from lc_amd.dense import dense_front_end
    from oracle import dense_oracle

    g = torch.Generator().manual_seed(B * H + W)
    xyz = torch.randn(B, 3, H, W, generator=g)
    wl = torch.randn(B, 2, H, W, generator=g) * 2
    ws = torch.exp(torch.randn(B, 1, 1, 1, generator=g) * 0.3 + 3)
    ns = torch.rand(B, 3, generator=g) * 40 + 10
    dev = torch.device("cuda:0")
    leaves = [t.to(dev).requires_grad_(True) for t in (xyz, wl, ws)]
    p2, s, p3 = dense_front_end(leaves[0], leaves[1], leaves[2], ns.to(dev), sample=sample, top_left=tl)
    l64 = [t.double().requires_grad_(True) for t in (xyz, wl, ws)]
    q2, r, q3 = dense_oracle.dense_front_end(l64[0], l64[1], l64[2], ns.double(), sample, tl)
    assert p2.shape == q2.shape and torch.equal(p2.cpu().double(), q2)
    assert rel_err(s.detach().cpu(), r.detach()) <= 2e-6 and rel_err(p3.detach().cpu(), q3.detach()) <= 1e-6
    cs, c3 = torch.randn(s.shape, generator=g), torch.randn(p3.shape, generator=g)
    gk = torch.autograd.grad([s, p3], leaves, [cs.to(dev), c3.to(dev)])
    go = torch.autograd.grad([r, q3], l64, [cs.double(), c3.double()])
    for a, b_ in zip(gk, go):
        assert rel_err(a.cpu(), b_) <= 5e-6


def test_dense_loss_fn_end_to_end_on_gpu():
    """lc_amd.losses.Loss_fn dense branch with fused front end + fused loss == golden trajectory of the reference."""
    import os
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import run
    from tests.util import GOLDEN

    z = np.load(os.path.join(GOLDEN, "lossfn_dense_f64.npz"))
    rec = run(Loss_fn, "dense", list(z["steps"]), torch.float32, device=torch.device("cuda:0"))
    for k in z.files:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), k
        elif "_grad_" in k:
            assert rel_err(rec[k], z[k]) <= 2e-3, k
        else:
            assert rel_err(rec[k], z[k]) <= 1e-3, k


def test_sparse_loss_fn_end_to_end_on_gpu():
    import os
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import run
    from tests.util import GOLDEN

    z = np.load(os.path.join(GOLDEN, "lossfn_sparse_f64.npz"))
    rec = run(Loss_fn, "sparse", list(z["steps"]), torch.float32, device=torch.device("cuda:0"))
    for k in z.files:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), k
        else:
            assert rel_err(rec[k], z[k]) <= 2e-3, k


def test_bin_loss_fn_end_to_end_on_gpu():
    """ZebraPose structure: binary-code decode kernels + fused front end + fused loss vs the reference trajectory."""
    import os
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import run
    from tests.util import GOLDEN

    z = np.load(os.path.join(GOLDEN, "lossfn_bin_f64.npz"))
    rec = run(Loss_fn, "bin", list(z["steps"]), torch.float32, device=torch.device("cuda:0"))
    for k in z.files:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), k
        elif "_grad_" in k:
            assert rel_err(rec[k], z[k]) <= 2e-3, k
        else:
            assert rel_err(rec[k], z[k]) <= 1e-3, k


def _train_shape_record(kind, head_dtype=None, loss_scale=1.0, round_heads_to=None):
    import os
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import TRAIN_KINDS, run
    from tests.util import GOLDEN

    z = np.load(os.path.join(GOLDEN, f"lossfn_{kind}.npz"))
    assert list(z["steps"]) == TRAIN_KINDS[kind][0]
    rec = run(Loss_fn, kind, list(z["steps"]), torch.float32, device=torch.device("cuda:0"), head_dtype=head_dtype, loss_scale=loss_scale, round_heads_to=round_heads_to)
    assert set(rec) == {k for k in z.files if not k.startswith("f32_")}
    return z, rec


@pytest.mark.parametrize("kind", ["dense_glmo", "bin_zlmo", "sparse_metric", "dense_plumb"])
def test_loss_fn_at_the_reference_training_shapes(kind):
    """`Loss_fn.forward` + backward at the shapes the reference's training configs execute against trajectories of the
    UNMODIFIED reference class in float64 (`tests/golden/lossfn_<kind>.npz`, `gen_golden_lossfn.py --train-shapes`):
    dense_glmo -- B=4, 64x64 maps, stride 2 => N=1024: the many-workgroups-per-sample front end and the TILED loss kernel with its cached
    workspace inside autograd, NormClipper hooks over the weight-logit gradient maps, aux losses, warm-up blend (configs/glmo.yaml:72-79);
    bin_zlmo -- B=4, 128x128 maps, stride 3 => N=1849, 7+7+7 code planes decoded with ground-truth bits, model transform, L1 segmentation,
    `Loss_xyz_bin` with its EMA histogram (configs/zlmo.yaml:74-83);  sparse_metric -- B=256, N=64 keypoints (gsplmo's loss block);
    dense_plumb -- BASELINE configs[0]'s shape: B=16, 32x32 maps, stride 2 => N=256, the LAST size the one-workgroup loss kernel takes
    (four correspondences per lane; `lc_loss.hip` switches to the tiled kernel above it).
    Same tolerances as the 16x16 trajectories above: 1e-4 on every loss, 2e-3 of a gradient map's largest entry, 1e-3 on the states."""
    z, rec = _train_shape_record(kind)
    for k in rec:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), (k, float(rec[k]), float(z[k]))
        elif "_grad_" in k:
            assert np.isfinite(rec[k]).all() and rel_err(rec[k], z[k]) <= 2e-3, (k, rel_err(rec[k], z[k]))
        else:
            assert rel_err(rec[k], z[k]) <= 1e-3, (k, rec[k], z[k])
    if kind != "sparse_metric":  # the trajectory did exercise the clipper: max_norm moved at every call
        mn = [float(rec[f"s{i}_state_weight_grad_clipper.max_norm"]) for i in range(len(z["steps"]))]
        assert all(m > 0 for m in mn) and len(set(mn)) == len(mn)


@pytest.mark.parametrize("head_dtype,loss_scale,tol", [
    (torch.float16, 4096.0, dict(loss=2e-3, grad=3e-2, grad_xyz_noc=3e-2, state=5e-3, twin_loss=1e-4, twin_grad=2e-3)),
    (torch.bfloat16, 1.0, dict(loss=2e-2, grad=6e-2, grad_xyz_noc=3e-1, state=5e-2, twin_loss=1e-4, twin_grad=1.2e-2))], ids=["fp16", "bf16"])
@pytest.mark.parametrize("kind", ["dense_glmo", "bin_zlmo"])
def test_loss_fn_at_training_shapes_with_half_precision_heads(kind, head_dtype, loss_scale, tol):
    """BASELINE configs[2] / [4]: a mixed-precision backbone hands fp16 / bf16 maps to `Loss_fn`.  Two comparisons per step:

    (1) against the SAME reference trajectories as the fp32 test (float64, full-precision inputs), at what rounding the INPUTS to 11 / 8
    significant bits does to each quantity: losses 2e-3 / 2e-2, gradient maps 3e-2 / 6e-2 of their largest entry, max_norm and the code
    histogram 5e-3 / 5e-2.  The gradient of the xyz head under bf16 gets 0.3: its largest entries are the LC loss's, proportional to a
    pixel's reprojection error (~0.2 px here), and a bf16 coordinate is off by up to 0.09 mm = 0.04 px -- a property of the inputs, not of
    the kernels (measured 0.19; fp16, 8x finer: 0.015).  Excluded, and counted: the sign of `loss_noc`'s L1 gradient (losses.py:281-283)
    at pixels whose coordinate error is smaller than the rounding of the coordinate.
    (2) against the fp32 HIP step on the same ROUNDED values (itself pinned to the reference by the test above): what is left is the
    16-bit write of a map's gradient -- 2e-3 / 1.2e-2 of the largest entry, losses 1e-4.

    fp16 runs under a loss scale of 4096, as a GradScaler-driven step does (gradients of 5e-8 at the start of the warm-up ramp are below
    fp16's subnormal step otherwise); gradients and max_norm are compared after un-scaling."""
    from lc_amd import synth

    z, rec = _train_shape_record(kind, head_dtype, loss_scale)
    _, twin = _train_shape_record(kind, None, loss_scale, round_heads_to=head_dtype)
    for k in rec:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= tol["loss"] * max(1.0, abs(float(z[k]))), (k, float(rec[k]), float(z[k]))
            assert abs(float(rec[k]) - float(twin[k])) <= tol["twin_loss"] * max(1.0, abs(float(twin[k]))), (k, float(rec[k]), float(twin[k]))
        elif "_grad_" in k:
            got, want = np.asarray(rec[k], np.float64), np.asarray(z[k], np.float64)
            assert np.isfinite(got).all()
            assert rel_err(got, twin[k]) <= tol["twin_grad"], (k, "vs the fp32 step on the rounded values", rel_err(got, twin[k]))
            if k.endswith("_grad_xyz_noc"):
                gt, out = synth.train_inputs(kind, seed=int(k[1:k.index("_")]))
                m, x, t = gt["msk_noc"][:, None].float(), out["xyz_noc"], gt["xyz_noc_tgt"]
                flips = (torch.sign(x * m - t) != torch.sign(x.to(head_dtype).float() * m - t)).numpy()
                assert flips.sum() <= 0.05 * float(m.sum()) * 3, (k, int(flips.sum()))
                got, want = np.where(flips, 0.0, got), np.where(flips, 0.0, want)
            err = np.abs(got - want).max() / np.abs(want).max()
            assert err <= tol["grad_xyz_noc" if k.endswith("_grad_xyz_noc") else "grad"], (k, err)
        else:
            assert rel_err(rec[k], z[k]) <= tol["state"], (k, rec[k], z[k])


@pytest.mark.parametrize("thr", [0.5, 0.3, 0.9])
def test_front_end_visibility_mask_equals_torch(thr):
    """dense_front_end_with_visibility: the mask of the sampled pixels from the front-end launch is torch's
    `sigmoid(logits) > thr` on the stride slice (test.py:88-90) bit for bit, and the other outputs are those of dense_front_end."""
    from lc_amd.dense import dense_front_end, dense_front_end_with_visibility

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(int(thr * 100))
    B, H, W, s = 5, 48, 40, 2
    xyz, wl, ws = torch.randn(B, 3, H, W, generator=g).to(dev), torch.randn(B, 2, H, W, generator=g).to(dev), torch.rand(B, 1, 1, 1, generator=g).to(dev) + 0.5
    ns = torch.rand(B, 3, generator=g).to(dev) + 0.5
    # logits around the decision boundary logit(thr), incl. exact ties of the fp32 sigmoid
    base = float(np.log(thr / (1 - thr)))
    vl = (base + torch.randn(B, 1, H, W, generator=g) * 1e-3).to(dev)
    vl[0, 0, :4] = base
    p2, w2, x3, vis = dense_front_end_with_visibility(xyz, wl, ws, ns, vl, thr, sample=s)
    q2, v2, y3 = dense_front_end(xyz, wl, ws, ns, sample=s, top_left=(0, 0))
    assert torch.equal(p2, q2) and torch.equal(w2, v2) and torch.equal(x3, y3)
    want = (torch.sigmoid(vl) > thr).squeeze(1)[..., 0::s, 0::s].flatten(-2)
    assert vis.dtype == torch.bool and torch.equal(vis, want) and 0 < int(want.sum()) < want.numel()


@pytest.mark.parametrize("seg_type", ["bce", "l1"])
@pytest.mark.parametrize("B,H,W,with_xyz,with_w,mask_dtype", [(32, 64, 64, True, True, torch.bool), (3, 17, 23, True, False, torch.float32),
                                                               (5, 32, 32, False, True, torch.bool), (2, 128, 128, True, True, torch.uint8)])
def test_dense_aux_losses_equal_the_torch_formulas(B, H, W, with_xyz, with_w, mask_dtype, seg_type):
    """lc_dense_aux_{fwd,bwd}_f32 (loss_noc, loss_seg, loss_weight_seg of losses.py:281-316 in one launch each way) against the
    reference's torch formulas in float64: values to 2e-6, gradients to 2e-6 of their largest entry, with non-trivial upstream
    cotangents, exact zeros of the L1 difference (sign(0) = 0) and saturated logits."""
    import torch.nn.functional as F

    from lc_amd.dense_aux import dense_aux_losses
    from lc_amd.losses import Loss_seg_L1

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * H + W)
    xyz = torch.randn(B, 3, H, W, generator=g) if with_xyz else None
    msk = torch.rand(B, H, W, generator=g) > 0.4
    tgt = torch.randn(B, 3, H, W, generator=g) * msk[:, None] if with_xyz else None
    if with_xyz:
        xyz[0, :, :2] = tgt[0, :, :2]  # exact zeros of the difference where the mask is set
    seg = torch.randn(B, 1, H, W, generator=g) * 4
    seg[0, 0, 0, :4] = torch.tensor([60.0, -60.0, 0.0, 100.0])[:min(4, W)]
    vis = (torch.rand(B, H, W, generator=g) > 0.5).float()
    wl = torch.randn(B, 2, H, W, generator=g) * 2 if with_w else None
    up = torch.tensor([0.7, 1.3, 0.4])

    def leaves(dt, d):
        return [None if t is None else t.to(device=d, dtype=dt).requires_grad_(True) for t in (xyz, seg, wl)]

    # reference formulas, float64 on the CPU
    x64, s64, w64 = leaves(torch.float64, "cpu")
    seg_fn = F.binary_cross_entropy_with_logits if seg_type == "bce" else Loss_seg_L1()
    want = [F.l1_loss(x64 * msk[:, None], tgt.double()) if with_xyz else None, seg_fn(s64, vis[:, None].double(), reduction="mean"),
            seg_fn(w64, vis[:, None].double().expand_as(w64), reduction="mean") if with_w else None]
    sum(u * v for u, v in zip(up.double(), want) if v is not None).backward()
    # fused launches
    xg, sg, wg = leaves(torch.float32, dev)
    m = msk.to(dev) if mask_dtype == torch.bool else msk.to(device=dev, dtype=mask_dtype)
    got = dense_aux_losses(xg, m if with_xyz else None, tgt.to(dev) if with_xyz else None, sg, vis.to(dev), wg, seg_type)
    sum(u * v for u, v, w_ in zip(up.to(dev), got, want) if w_ is not None).backward()
    for a, b in zip(got, want):
        if b is not None:
            assert abs(float(a) - float(b)) <= 2e-6 * max(1.0, abs(float(b))), (float(a), float(b))
    for a, b in ((xg, x64), (sg, s64), (wg, w64)):
        if b is not None:
            assert (a.grad.cpu().double() - b.grad).abs().max() <= 2e-6 * b.grad.abs().max(), float((a.grad.cpu().double() - b.grad).abs().max())


@pytest.mark.parametrize("B,C,H,W", [(32, 17, 64, 64), (3, 21, 37, 29), (2, 72, 16, 16), (4, 5, 128, 128), (32, 21, 128, 128), (5, 7, 6, 6), (3, 4, 40, 36)])
def test_xyz_bin_loss_equals_the_torch_formulas(B, C, H, W):
    """lc_xyz_bin_loss_fwd2 / _bwd2 against Loss_xyz_bin's torch formulas (losses.py:196-216) over three steps of the EMA histogram -- every launch
    form: planes of 16-byte requests (64x64, 128x128; zlmo's B=32 x 21 x 128 x 128: two units per workgroup and channel), planes that do not fill a
    workgroup's share (6x6, 40x36), the element-wise walk (37x29):
    loss to 2e-6, histogram to 1e-6, gradient to 2e-6 of its largest entry (float64 torch on the CPU as the reference)."""
    from lc_amd.losses import Loss_xyz_bin

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C * H)
    fused, plain = Loss_xyz_bin(C).to(dev), Loss_xyz_bin(C).double()
    for step in range(3):
        logits = torch.randn(B, C, H, W, generator=g) * 3
        logits[0, 0, 0, :3] = torch.tensor([0.0, 50.0, -50.0])[:min(3, W)]
        bits = (logits > 0) ^ (torch.rand(B, C, H, W, generator=g) < 0.2)  # a head that is right four times out of five
        vis = torch.randn(B, 1, H, W, generator=g)
        a = logits.to(dev).requires_grad_(True)
        b = logits.double().requires_grad_(True)
        la, lb = fused(a, bits.to(dev), vis.to(dev)), plain(b, bits, vis.double())
        (la * 1.7).backward()
        (lb * 1.7).backward()
        assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb))), (step, float(la), float(lb))
        assert (fused.histogram.cpu().double() - plain.histogram).abs().max() <= 1e-6
        assert (a.grad.cpu().double() - b.grad).abs().max() <= 2e-6 * b.grad.abs().max()
    assert float((fused.histogram - 0.5).abs().max()) > 1e-3  # the EMA moved


@pytest.mark.parametrize("vis_sign", [-1.0, 1.0])
def test_xyz_bin_loss_with_an_empty_and_a_full_visibility_mask(vis_sign):
    """Edge cases of Loss_xyz_bin (losses.py:203-216): no visible pixel at all (the histogram update divides by `msk_hard.sum() + 1`
    = 1, every masked logit is 0 so each pixel costs log 2) and every pixel visible; same tolerances as the seeded cases."""
    from lc_amd.losses import Loss_xyz_bin

    dev = torch.device("cuda:0")
    B, C, H, W = 3, 9, 24, 40
    g = torch.Generator().manual_seed(7)
    fused, plain = Loss_xyz_bin(C).to(dev), Loss_xyz_bin(C).double()
    for step in range(2):
        logits = torch.randn(B, C, H, W, generator=g) * 3
        bits = torch.rand(B, C, H, W, generator=g) < 0.5
        vis = vis_sign * (torch.rand(B, 1, H, W, generator=g) + 0.1)
        a, b = logits.to(dev).requires_grad_(True), logits.double().requires_grad_(True)
        la, lb = fused(a, bits.to(dev), vis.to(dev)), plain(b, bits, vis.double())
        la.backward()
        lb.backward()
        assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb))), (step, float(la), float(lb))
        assert (fused.histogram.cpu().double() - plain.histogram).abs().max() <= 1e-6
        assert (a.grad.cpu().double() - b.grad).abs().max() <= 2e-6 * max(float(b.grad.abs().max()), 1e-30)
        if vis_sign < 0:
            assert float(a.grad.abs().max()) == 0.0 and abs(float(la) - 0.6931471805599453) < 1e-6


def test_dense_aux_losses_with_an_empty_mask_and_one_pixel_maps():
    """Edge cases of the dense auxiliary losses (losses.py:281-316): an all-false foreground mask (loss_noc = mean |0 - tgt|, zero
    coordinate gradient) and 1x1 maps (one pixel per sample: the vectorised path must fall back to scalars)."""
    import torch.nn.functional as F

    from lc_amd.dense_aux import dense_aux_losses

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    for B, H, W, empty in ((4, 16, 16, True), (5, 1, 1, False), (2, 3, 5, True)):
        xyz, tgt = torch.randn(B, 3, H, W, generator=g), torch.randn(B, 3, H, W, generator=g)
        msk = torch.zeros(B, H, W, dtype=torch.bool) if empty else torch.ones(B, H, W, dtype=torch.bool)
        seg, wl = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 2, H, W, generator=g)
        vis = (torch.rand(B, H, W, generator=g) > 0.5).float()
        x64, s64, w64 = (t.double().requires_grad_(True) for t in (xyz, seg, wl))
        want = [F.l1_loss(x64 * msk[:, None], tgt.double()), F.binary_cross_entropy_with_logits(s64, vis[:, None].double()),
                F.binary_cross_entropy_with_logits(w64, vis[:, None].double().expand_as(w64))]
        sum(want).backward()
        xg, sg, wg = (t.to(dev).requires_grad_(True) for t in (xyz, seg, wl))
        got = dense_aux_losses(xg, msk.to(dev), tgt.to(dev), sg, vis.to(dev), wg, "bce")
        sum(got).backward()
        for a, b in zip(got, want):
            assert abs(float(a) - float(b)) <= 2e-6 * max(1.0, abs(float(b))), (B, H, W, float(a), float(b))
        for a, b in ((xg, x64), (sg, s64), (wg, w64)):
            assert (a.grad.cpu().double() - b.grad).abs().max() <= 2e-6 * max(float(b.grad.abs().max()), 1e-30)
        if empty:
            assert float(xg.grad.abs().max()) == 0.0
