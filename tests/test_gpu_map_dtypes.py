"""fp16 / bf16 network outputs are read NATIVELY by the f1 / f3 kernels (dense front end fwd / bwd, front end + selection, code decode fwd /
bwd, the auxiliary losses): no up-cast copy, fp32 arithmetic, gradients of a map written in the map's own type -- and a head that is a
channel slice of the network's (B,C_all,H,W) output (`ptnet.py:56`) is consumed where it lies, no `.contiguous()` copy.

The bar: equality with the fp32 kernel on the same rounded values -- forward outputs bit for bit (every reduction adds the
same values in the same order whatever the map type), gradient maps equal to the fp32 gradient rounded to nearest even into the map's type
(asserted EXACTLY, tighter than the 1 ulp asked for) -- and zero `aten::_to_copy` launches on the way (torch profiler)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]


def _ops(fn):
    """Names of the aten ops `fn` dispatches (CPU-side profiler events: one per op call, whatever the device does with it)."""
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = fn()
        torch.cuda.synchronize()
    return out, [e.name for e in prof.events()]


def _no_copies(names):
    bad = [n for n in names if n in ("aten::_to_copy", "aten::clone", "aten::copy_")]  # (`aten::to` / `aten::contiguous` that do nothing dispatch neither)
    assert not bad, bad


def _heads(B, H, W, dtype, seed, sliced):
    """xyz (B,3,H,W), weight logits (B,2,H,W), visibility logits (B,1,H,W) of `dtype`; sliced: channel slices of ONE (B,9,H,W) tensor (with
    unrelated channels between them), as the reference's network hands them over."""
    g = torch.Generator().manual_seed(seed)
    raw = (torch.randn(B, 9, H, W, generator=g) * torch.tensor([1, 1, 1, 9, 2, 2, 9, 9, 3.0]).view(1, 9, 1, 1)).to(DEV).to(dtype)
    xyz, wl, vl = raw[:, 0:3], raw[:, 4:6], raw[:, 8:9]
    if not sliced:
        xyz, wl, vl = xyz.contiguous(), wl.contiguous(), vl.contiguous()
    ws = (torch.rand(B, generator=g) * 30 + 5).to(DEV)
    ns = (torch.rand(B, 3, generator=g) * 50 + 20).to(DEV)
    return xyz, wl, vl, ws, ns


@pytest.mark.parametrize("sliced", [False, True], ids=["dense", "channel_slices"])
@pytest.mark.parametrize("dtype", DTYPES + [torch.float32])
@pytest.mark.parametrize("B,H,W,sample,top_left", [(5, 32, 32, 2, (1, 0)), (3, 37, 45, 3, (2, 1)), (64, 128, 128, 1, (0, 0)),
                                                  (3, 6, 7, 1, (0, 0))])  # H*W % 4 == 2: odd samples of a slice start off the four-element boundary
def test_dense_front_end_reads_maps_natively(B, H, W, sample, top_left, dtype, sliced):
    from lc_amd.dense import dense_front_end

    xyz, wl, _, ws, ns = _heads(B, H, W, dtype, seed=B + H, sliced=sliced)
    xyz_h, wl_h = xyz.detach().requires_grad_(True), wl.detach().requires_grad_(True)
    ws_h = ws.clone().requires_grad_(True)
    (u, s, x), names = _ops(lambda: dense_front_end(xyz_h, wl_h, ws_h, ns, sample=sample, top_left=top_left))
    _no_copies(names)
    g = torch.Generator().manual_seed(1)
    gs, gx = torch.randn(s.shape, generator=g).to(DEV), torch.randn(x.shape, generator=g).to(DEV)
    (d_xyz, d_wl, d_ws), names = _ops(lambda: torch.autograd.grad((s * gs).sum() + (x * gx).sum(), (xyz_h, wl_h, ws_h)))
    assert "aten::_to_copy" not in names
    # the fp32 kernel on the same (rounded) values, dense
    xyz_f, wl_f = xyz.float().contiguous().requires_grad_(True), wl.float().contiguous().requires_grad_(True)
    ws_f = ws.clone().requires_grad_(True)
    u2, s2, x2 = dense_front_end(xyz_f, wl_f, ws_f, ns, sample=sample, top_left=top_left)
    r_xyz, r_wl, r_ws = torch.autograd.grad((s2 * gs).sum() + (x2 * gx).sum(), (xyz_f, wl_f, ws_f))
    assert torch.equal(u, u2) and torch.equal(s, s2) and torch.equal(x, x2)  # forward: bit for bit
    assert d_xyz.dtype == dtype and d_wl.dtype == dtype and d_ws.dtype == torch.float32
    assert torch.equal(d_xyz, r_xyz.to(dtype)) and torch.equal(d_wl, r_wl.to(dtype)) and torch.equal(d_ws, r_ws)


@pytest.mark.parametrize("sliced", [False, True], ids=["dense", "channel_slices"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["mask", "quantile", "quantile_in_mask"])
@pytest.mark.parametrize("B,H,W,sample", [(6, 64, 64, 2), (4, 128, 128, 1), (3, 37, 45, 1)])
def test_front_end_and_selection_read_maps_natively(B, H, W, sample, mode, dtype, sliced):
    from lc_amd.dense import dense_front_end_select, dense_front_end_with_visibility

    xyz, wl, vl, ws, ns = _heads(B, H, W, dtype, seed=H, sliced=sliced)
    kw = dict(seg_thresh=0.5, sample=sample, quantile=0.3, min_count=4, seed=3)
    got, names = _ops(lambda: dense_front_end_select(xyz, wl, ws, ns, vl, mode, **kw))
    _no_copies(names)
    want = dense_front_end_select(xyz.float().contiguous(), wl.float().contiguous(), ws, ns, vl.float().contiguous(), mode, **kw)
    cnt = want[3]
    assert torch.equal(got[3], cnt)
    N = got[0].shape[1]
    live = torch.arange(N, device=DEV)[None, :] < cnt[:, None]
    for k in (0, 1, 2, 4):
        m = live if got[k].dim() == 2 else live[..., None].expand_as(got[k])
        assert torch.equal(got[k][m], want[k][m]), k
    rows, names = _ops(lambda: dense_front_end_with_visibility(xyz, wl, ws, ns, vl, 0.5, sample=sample))
    _no_copies(names)
    rows_f = dense_front_end_with_visibility(xyz.float().contiguous(), wl.float().contiguous(), ws, ns, vl.float().contiguous(), 0.5, sample=sample)
    assert all(torch.equal(a, b) for a, b in zip(rows, rows_f))


@pytest.mark.parametrize("sliced", [False, True], ids=["dense", "channel_slice"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,bits,sample,top_left", [(4, 32, 32, (6, 6, 5), 2, (1, 1)), (3, 128, 128, (7, 7, 7), 1, (0, 0)), (2, 17, 23, (5, 7, 3), 3, (0, 2))])
def test_code_decode_reads_logits_natively(B, H, W, bits, sample, top_left, dtype, sliced):
    from lc_amd import floatbits

    g = torch.Generator().manual_seed(H)
    C = sum(bits)
    raw = (torch.randn(B, C + 5, H, W, generator=g) * 2).to(DEV).to(dtype)
    lg = raw[:, 2:2 + C] if sliced else raw[:, 2:2 + C].contiguous()
    lf = lg.float().contiguous()
    gt_bits = (torch.rand(B, C, H, W, generator=g) > 0.5).to(DEV)
    gt_msk = (torch.rand(B, H, W, generator=g) > 0.3).to(DEV)
    ns = (torch.rand(B, 3, generator=g) * 50 + 20).to(DEV)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, :3] = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    T[:, :3, 3] = torch.randn(B, 3, generator=g)
    T = T.to(DEV)
    # inference decode, both output layouts
    (noc, planes), names = _ops(lambda: (floatbits.nn_logits2noc(lg, list(bits)), floatbits.nn_logits2xyz_planes(lg, list(bits), ns, T)))
    _no_copies(names)
    assert torch.equal(noc, floatbits.nn_logits2noc(lf, list(bits))) and torch.equal(planes, floatbits.nn_logits2xyz_planes(lf, list(bits), ns, T))
    # training decode on the strided subset, forward and backward
    lh = lg.detach().requires_grad_(True)
    out, names = _ops(lambda: floatbits.decode_with_gt_strided(lh, gt_bits, list(bits), gt_msk, sample=sample, top_left=top_left, out_scale=ns, out_xform=T))
    _no_copies(names)
    go = torch.randn(out.shape, generator=g).to(DEV)
    (d,), names = _ops(lambda: torch.autograd.grad((out * go).sum(), (lh,)))
    assert "aten::_to_copy" not in names
    l32 = lf.clone().requires_grad_(True)
    out32 = floatbits.decode_with_gt_strided(l32, gt_bits, list(bits), gt_msk, sample=sample, top_left=top_left, out_scale=ns, out_xform=T)
    (d32,) = torch.autograd.grad((out32 * go).sum(), (l32,))
    assert torch.equal(out, out32) and d.dtype == dtype and torch.equal(d, d32.to(dtype))


@pytest.mark.parametrize("sliced", [False, True], ids=["dense", "channel_slices"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("seg", ["bce", "l1"])
@pytest.mark.parametrize("B,H,W", [(4, 32, 32), (3, 17, 23), (32, 64, 64)])
def test_auxiliary_losses_read_maps_natively(B, H, W, seg, dtype, sliced):
    from lc_amd import dense_aux

    xyz, wl, vl, _, _ = _heads(B, H, W, dtype, seed=W, sliced=sliced)
    g = torch.Generator().manual_seed(2)
    msk_noc = (torch.rand(B, H, W, generator=g) > 0.4).to(DEV)
    msk_vis = (torch.rand(B, H, W, generator=g) > 0.4).float().to(DEV)
    tgt = torch.randn(B, 3, H, W, generator=g).to(DEV) * msk_noc[:, None]
    hs = [t.detach().requires_grad_(True) for t in (xyz, vl, wl)]
    fs = [t.float().contiguous().requires_grad_(True) for t in (xyz, vl, wl)]
    losses, names = _ops(lambda: dense_aux.dense_aux_losses(hs[0], msk_noc, tgt, hs[1], msk_vis, hs[2], seg))
    _no_copies(names)
    want = dense_aux.dense_aux_losses(fs[0], msk_noc, tgt, fs[1], msk_vis, fs[2], seg)
    assert all(torch.equal(a, b) for a, b in zip(losses, want))
    w = torch.tensor([0.7, 1.3, 0.4], device=DEV)
    grads, names = _ops(lambda: torch.autograd.grad(sum(a * b for a, b in zip(losses, w)), hs))
    assert "aten::_to_copy" not in names
    ref = torch.autograd.grad(sum(a * b for a, b in zip(want, w)), fs)
    for a, b in zip(grads, ref):
        assert a.dtype == dtype and torch.equal(a, b.to(dtype))
    # Loss_xyz_bin on code logits of the same type
    C = 17
    raw = (torch.randn(B, C + 3, H, W, generator=g) * 2).to(DEV).to(dtype)
    lg = raw[:, 1:1 + C] if sliced else raw[:, 1:1 + C].contiguous()
    gt_bits = (torch.rand(B, C, H, W, generator=g) > 0.5).to(DEV)
    h1, h2 = torch.full((C,), 0.5, device=DEV), torch.full((C,), 0.5, device=DEV)
    lh, lf = lg.detach().requires_grad_(True), lg.float().contiguous().requires_grad_(True)
    loss, names = _ops(lambda: dense_aux.xyz_bin_loss(lh, gt_bits, vl, h1, 0.05))
    _no_copies(names)
    loss32 = dense_aux.xyz_bin_loss(lf, gt_bits, vl.float().contiguous(), h2, 0.05)
    assert torch.equal(loss, loss32) and torch.equal(h1, h2)
    (d,), (d32,) = torch.autograd.grad(loss * 1.7, (lh,)), torch.autograd.grad(loss32 * 1.7, (lf,))
    assert d.dtype == dtype and torch.equal(d, d32.to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["zlmo", "glmo"])
def test_test_time_path_on_16bit_network_outputs(name, dtype):
    """`solve_pnp` at the reference's test-time configs on fp16 / bf16 head outputs handed over as channel slices: no cast, no copy, and the
    poses of the fp32 path on the same rounded values bit for bit."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp

    cfg, gt, out = synth.test_time_inputs(name, B=8, seed=4)
    cfg = AttrDict(cfg)
    gt = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
    keys = [k for k in ("xyz_noc", "xyz_noc_bin", "xyz_weight_logits", "msk_vis_logits") if k in out]
    raw = torch.cat([out[k] for k in keys], 1).to(DEV).to(dtype)  # the network's one output tensor
    half, c0 = {"xyz_weights_scale": out["xyz_weights_scale"].to(DEV)}, 0
    for k in keys:
        half[k] = raw[:, c0:c0 + out[k].shape[1]]
        c0 += out[k].shape[1]
    got, names = _ops(lambda: solve_pnp(cfg, half, gt))
    _no_copies(names)
    want = solve_pnp(cfg, {k: (v.float().contiguous() if k in keys else v) for k, v in half.items()}, gt)
    assert list(got) == list(want) and all(torch.equal(got[k], want[k]) for k in got)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fp32_coordinate_map_next_to_16bit_logits_keeps_an_fp32_gradient(dtype):
    """the differentiable front end with an fp32 `xyz_noc` next to 16-bit weight logits.  The coordinate gradient is the
    scatter of `g_pts3d * noc_scale` -- it does not involve the logits -- so it must come back in fp32 and bit-identical to the all-fp32
    call, tiny values included (written in fp16 they would flush to zero); the logits' gradient stays in the logits' type."""
    from lc_amd.dense import dense_front_end

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, H, W, s = 3, 32, 36, 2
    xyz = torch.randn(B, 3, H, W, generator=g).to(dev)
    wl = (torch.randn(B, 2, H, W, generator=g) * 2).to(dev)
    ws = (torch.rand(B, 1, 1, 1, generator=g) + 0.5).to(dev)
    ns = (torch.rand(B, 3, generator=g) + 0.5).to(dev)
    cot = (torch.randn(B, (H // s) * (W // s), 3, generator=g) * 1e-9).to(dev)  # far below fp16's smallest subnormal (6e-8)
    cs = torch.randn(B, (H // s) * (W // s), 2, generator=g).to(dev)
    grads = {}
    for name, wl_t in (("mixed", wl.to(dtype)), ("fp32", wl.to(dtype).float())):
        x, l = xyz.clone().requires_grad_(True), wl_t.clone().requires_grad_(True)
        _, inv_std, p3 = dense_front_end(x, l, ws, ns, sample=s, top_left=(1, 0))
        gx, gl = torch.autograd.grad([p3, inv_std], [x, l], [cot, cs])
        grads[name] = (gx, gl)
    gx, gl = grads["mixed"]
    assert gx.dtype == torch.float32 and gl.dtype == dtype
    assert torch.equal(gx, grads["fp32"][0]) and float(gx.abs().max()) > 0 and float(gx.abs().max()) < 6e-8
    assert (gl.float() - grads["fp32"][1]).abs().max() <= (8e-3 if dtype == torch.bfloat16 else 1e-3) * grads["fp32"][1].abs().max()
