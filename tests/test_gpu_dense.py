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
