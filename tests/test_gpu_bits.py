"""GPU parity of the binary-code decode kernels against goldens from the reference's floatbits.py."""
import numpy as np
import pytest
import torch

from tests.util import golden_files, case_name, rel_err

pytestmark = pytest.mark.gpu
FILES = golden_files("bits_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "bits_") for p in FILES])
def test_bits_decode_vs_reference(path):
    from lc_amd import floatbits as fb

    z = np.load(path)
    dev = torch.device("cuda:0")
    bits = [int(b) for b in z["bits"]]
    lg = torch.from_numpy(z["in_logits"]).to(dev).requires_grad_(True)
    raw = torch.from_numpy(z["in_raw_bits"]).to(dev)
    msk = torch.from_numpy(z["in_msk"]).to(dev)
    out = fb.nn_logits2noc_with_gt(lg, raw, bits, msk)
    (gl,) = torch.autograd.grad(out, lg, torch.from_numpy(z["in_ct"]).to(dev))
    assert rel_err(out.detach().cpu(), z["f64_noc_gt"]) <= 2e-6
    assert rel_err(gl.cpu(), z["f64_g_logits"]) <= 5e-6
    inf = fb.nn_logits2noc(lg.detach(), bits)
    assert rel_err(inf.cpu(), z["f64_noc_inf"]) <= 2e-6


def test_bits_strided_subset_and_zlmo_shape():
    """Sub-sample first, decode second (losses.py:163-184) on a 128x128 map with the zlmo bit budget."""
    from lc_amd import floatbits as fb
    from oracle import floatbits_oracle as orc

    g = torch.Generator().manual_seed(2)
    B, H, W, bits = 2, 128, 128, [7, 7, 6]
    noc = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    mod, raw = fb.nn_noc2target(noc, bits)
    lg = (mod.float() * 2 - 1) * (torch.rand(B, sum(bits), H, W, generator=g) * 3 + 0.1)
    lg = torch.where(torch.rand(lg.shape, generator=g) < 0.1, -lg, lg)
    msk = torch.rand(B, H, W, generator=g) > 0.2
    dev = torch.device("cuda:0")
    x = lg.to(dev).requires_grad_(True)
    out = fb.decode_with_gt_strided(x, raw.to(dev), bits, msk.to(dev), sample=3, top_left=(1, 2))
    x64 = lg.double().requires_grad_(True)
    ref = orc.nn_logits2noc_with_gt(x64[..., 1::3, 2::3], raw[..., 1::3, 2::3], bits, msk[..., 1::3, 2::3]).flatten(1, 2)
    assert out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) <= 2e-6
    ct = torch.randn(out.shape, generator=g)
    (gk,) = torch.autograd.grad(out, x, ct.to(dev))
    (go,) = torch.autograd.grad(ref, x64, ct.double())
    assert rel_err(gk.cpu(), go) <= 5e-6


@pytest.mark.parametrize("H,W,sample,tl", [(6, 10, 1, (0, 0)), (5, 7, 2, (1, 0)), (8, 12, 1, (0, 0)), (8, 12, 2, (0, 1))])
def test_bits_odd_widths_take_the_scalar_path(H, W, sample, tl):
    """Row lengths that are not a multiple of four (one pixel per thread) and the float4 path agree with the oracle."""
    from lc_amd import floatbits as fb
    from oracle import floatbits_oracle as orc

    g = torch.Generator().manual_seed(H * 100 + W)
    B, bits = 3, [5, 4, 3]
    noc = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    mod, raw = fb.nn_noc2target(noc, bits)
    lg = (mod.float() * 2 - 1) * (torch.rand(B, sum(bits), H, W, generator=g) * 3 + 0.1)
    lg = torch.where(torch.rand(lg.shape, generator=g) < 0.15, -lg, lg)
    msk = torch.rand(B, H, W, generator=g) > 0.3
    dev = torch.device("cuda:0")
    x = lg.to(dev).requires_grad_(True)
    out = fb.decode_with_gt_strided(x, raw.to(dev), bits, msk.to(dev), sample=sample, top_left=tl)
    x64 = lg.double().requires_grad_(True)
    sl = (Ellipsis, slice(tl[0], None, sample), slice(tl[1], None, sample))
    ref = orc.nn_logits2noc_with_gt(x64[sl], raw[sl], bits, msk[sl]).flatten(1, 2)
    assert out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) <= 2e-6
    ct = torch.randn(out.shape, generator=g)
    (gk,) = torch.autograd.grad(out, x, ct.to(dev))
    (go,) = torch.autograd.grad(ref, x64, ct.double())
    assert rel_err(gk.cpu(), go) <= 5e-6
    inf = fb.nn_logits2noc(lg.to(dev), bits)
    assert rel_err(inf.cpu(), orc.nn_logits2noc(lg.double(), bits)) <= 2e-6


@pytest.mark.parametrize("H,W,sample,tl,with_T", [(64, 64, 2, (1, 0), True), (32, 32, 1, (0, 0), True), (21, 30, 2, (0, 1), False), (16, 16, 1, (0, 0), False)])
def test_decode_with_the_coordinate_map_folded_in(H, W, sample, tl, with_T):
    """lc_bits_decode_gt_{fwd,bwd}2_f32: `noc * noc_scale` and the model transform `(xyz - T[:, :3, 3]) @ T[:, :3, :3]` of
    nn_out_to_xyz (losses.py:17-47) applied by the decode launches, against the plain decode followed by the torch ops (float64 on the
    oracle's decode): values and logit gradients."""
    from lc_amd import floatbits as fb
    from oracle import floatbits_oracle as orc

    g = torch.Generator().manual_seed(H + W)
    B, bits = 3, [6, 6, 5]
    noc = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    mod, raw = fb.nn_noc2target(noc, bits)
    lg = (mod.float() * 2 - 1) * (torch.rand(B, sum(bits), H, W, generator=g) * 3 + 0.1)
    lg = torch.where(torch.rand(lg.shape, generator=g) < 0.1, -lg, lg)
    msk = torch.rand(B, H, W, generator=g) > 0.2
    scale = torch.rand(B, 3, generator=g) * 100 + 20
    T = None
    if with_T:
        q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, :3, :3] = q
        T[:, :3, 3] = torch.randn(B, 3, generator=g) * 5
    dev = torch.device("cuda:0")
    x = lg.to(dev).requires_grad_(True)
    out = fb.decode_with_gt_strided(x, raw.to(dev), bits, msk.to(dev), sample=sample, top_left=tl, out_scale=scale.to(dev),
                                    out_xform=None if T is None else T.to(dev))
    x64 = lg.double().requires_grad_(True)
    sl = (Ellipsis, slice(tl[0], None, sample), slice(tl[1], None, sample))
    ref = orc.nn_logits2noc_with_gt(x64[sl], raw[sl], bits, msk[sl]).flatten(1, 2) * scale.double()[:, None]
    if T is not None:
        ref = (ref - T.double()[:, None, :3, 3]) @ T.double()[:, :3, :3]
    assert out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) <= 2e-6
    ct = torch.randn(out.shape, generator=g)
    (gk,) = torch.autograd.grad(out, x, ct.to(dev))
    (go,) = torch.autograd.grad(ref, x64, ct.double())
    assert rel_err(gk.cpu(), go) <= 5e-6


@pytest.mark.parametrize("H,W,with_T", [(64, 64, True), (30, 21, True), (128, 128, False)])
def test_inference_decode_straight_to_xyz_planes(H, W, with_T):
    """lc_bits_decode3 (Gray decode + noc_scale + model transform, written as (B,3,H,W) planes) against
    `nn_out_to_xyz(..., inference=True).permute(0, 3, 1, 2)`: the decode itself bit for bit (same kernel body), the coordinate map to
    fp32 rounding of a 3-term product sum."""
    from lc_amd import floatbits as fb
    from lc_amd.losses import nn_out_to_xyz

    g = torch.Generator().manual_seed(H * W)
    B, bits = 3, [7, 6, 6]
    dev = torch.device("cuda:0")
    lg = (torch.randn(B, sum(bits), H, W, generator=g) * 2).to(dev)
    scale = (torch.rand(B, 3, generator=g) * 100 + 20).to(dev)
    T = None
    if with_T:
        q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, :3, :3] = q
        T[:, :3, 3] = torch.randn(B, 3, generator=g) * 5
        T = T.to(dev)
    got = fb.nn_logits2xyz_planes(lg, bits, scale, T)
    want = nn_out_to_xyz(lg, scale, model_transform=T, bit_cnt=bits, inference=True).permute(0, 3, 1, 2)
    assert got.shape == want.shape == (B, 3, H, W) and got.is_contiguous()
    if T is None:
        assert torch.equal(got, want.contiguous())
    else:
        assert (got - want).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("H,W,sample,tl,bits,masked", [
    (128, 128, 3, (1, 2), [7, 7, 7], True),    # zlmo's training shape: tiles of 4 rows (42 sampled columns x 2 rows x 3 axes = 252 items)
    (128, 128, 3, (0, 0), [7, 7, 7], True),    # 43 sampled columns: tiles of 3 rows
    (128, 128, 3, (2, 1), [7, 7, 6], False),   # no object mask
    (64, 64, 2, (1, 1), [7, 7, 6], True),      # glmo-sized maps with binary heads, stride 2: tiles of 4 rows, two sampled rows each
    (30, 64, 2, (0, 1), [5, 4, 3], True),      # a height the tile rows do not divide
    (31, 32, 3, (2, 0), [6, 6, 6], True),
    (16, 128, 2, (1, 0), [9, 10, 3], True),    # axes of more than eight bits: a second round of requests
    (20, 24, 2, (0, 0), [5, 5, 5], True),      # 24 / 8 = 3 pieces per row does not divide the workgroup: the flat backward kernel
    (24, 40, 4, (3, 1), [5, 5, 5], True),      # stride 4: the flat backward kernel, the wide forward kernel
])
def test_strided_training_decode_launch_forms(H, W, sample, tl, bits, masked):
    """The strided-subset forms of round 6 (`lc_bits_decode_gt_fwd_wide_kernel`: every request of a pixel in flight at once;
    `lc_bits_decode_gt_bwd_tile_kernel<E, T, SAMPLE, R>`: zero rows first, one (pixel, axis) item per thread, every 16-byte piece written once) and
    the shapes that fall back to the flat kernels, against the float64 oracle (floatbits.py:130-160 restated) with the callers' coordinate map
    folded in; then fp16 / bf16 logits: the result of the fp32 kernels on the up-cast values, bit for bit (gradient: rounded once to the map's type)."""
    from lc_amd import floatbits as fb
    from oracle import floatbits_oracle as orc

    g = torch.Generator().manual_seed(H * 1000 + W * 10 + sample)
    B = 3
    noc = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    mod, raw = fb.nn_noc2target(noc, bits)
    lg = (mod.float() * 2 - 1) * (torch.rand(B, sum(bits), H, W, generator=g) * 3 + 0.1)
    lg = torch.where(torch.rand(lg.shape, generator=g) < 0.15, -lg, lg)
    msk = (torch.rand(B, H, W, generator=g) > 0.3) if masked else None
    scale = torch.rand(B, 3, generator=g) * 60 + 20
    ang = torch.rand(B, generator=g) * 0.6 - 0.3
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, 0, 0] = T[:, 1, 1] = ang.cos()
    T[:, 0, 1], T[:, 1, 0] = -ang.sin(), ang.sin()
    T[:, :3, 3] = torch.randn(B, 3, generator=g) * 2
    dev = torch.device("cuda:0")
    dm = None if msk is None else msk.to(dev)
    x = lg.to(dev).requires_grad_(True)
    out = fb.decode_with_gt_strided(x, raw.to(dev), bits, dm, sample=sample, top_left=tl, out_scale=scale.to(dev), out_xform=T.to(dev))
    x64 = lg.double().requires_grad_(True)
    sl = (Ellipsis, slice(tl[0], None, sample), slice(tl[1], None, sample))
    m64 = torch.ones(B, H, W, dtype=torch.bool) if msk is None else msk
    noc64 = orc.nn_logits2noc_with_gt(x64[sl], raw[sl], bits, m64[sl]).flatten(1, 2)
    ref = (noc64 * scale.double()[:, None] - T.double()[:, None, :3, 3]) @ T.double()[:, :3, :3]
    assert out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) <= 2e-6
    ct = torch.randn(out.shape, generator=g)
    (gk,) = torch.autograd.grad(out, x, ct.to(dev))
    (go,) = torch.autograd.grad(ref, x64, ct.double())
    assert rel_err(gk.cpu(), go) <= 5e-6
    off = torch.ones(H, W, dtype=torch.bool)
    off[tl[0]::sample, tl[1]::sample] = False
    assert float(gk[..., off.to(dev)].abs().max()) == 0.0  # nothing off the sampled pixels
    for dt in (torch.float16, torch.bfloat16):
        xh = lg.to(dev).to(dt).requires_grad_(True)
        xf = xh.detach().float().requires_grad_(True)
        oh = fb.decode_with_gt_strided(xh, raw.to(dev), bits, dm, sample=sample, top_left=tl, out_scale=scale.to(dev), out_xform=T.to(dev))
        of = fb.decode_with_gt_strided(xf, raw.to(dev), bits, dm, sample=sample, top_left=tl, out_scale=scale.to(dev), out_xform=T.to(dev))
        assert torch.equal(oh, of)
        (gh,), (gf,) = torch.autograd.grad(oh, xh, ct.to(dev)), torch.autograd.grad(of, xf, ct.to(dev))
        assert gh.dtype == dt and torch.equal(gh, gf.to(dt))
