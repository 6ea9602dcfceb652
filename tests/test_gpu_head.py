"""GPU parity of the fused spatial-softmax + soft-argmax kernels against golden vectors from ptnet.py (fp32 and fp64)."""
import numpy as np
import pytest
import torch

from tests.util import golden_files, case_name, rel_err

pytestmark = pytest.mark.gpu
FILES = golden_files("head_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "head_") for p in FILES])
def test_head_vs_golden(path):
    from lc_amd.ptnet import spatial_softargmax_2d_std, softargmax_2d_std

    z = np.load(path)
    dev = torch.device("cuda:0")
    lg = torch.from_numpy(z["in_logits"]).to(dev).requires_grad_(True)
    ctm, cts = torch.from_numpy(z["in_ct_mean"]).to(dev), torch.from_numpy(z["in_ct_std"]).to(dev)
    mean, std = spatial_softargmax_2d_std(lg)
    (gl,) = torch.autograd.grad([mean, std], [lg], [ctm, cts])
    # pixel-unit outputs in [0, 63]: 2e-4 px abs (fp32 reference itself sits ~1e-5 from fp64)
    assert (mean.detach().cpu().double() - torch.from_numpy(z["f64_mean"])).abs().max().item() <= 2e-4
    assert (std.detach().cpu().double() - torch.from_numpy(z["f64_std"])).abs().max().item() <= 2e-4
    assert rel_err(gl.cpu(), z["f64_g_logits"]) <= 2e-4
    assert rel_err(gl.cpu(), z["f32_g_logits"]) <= 2e-4
    # the function on its own (probabilities in), ptnet.py:100-115
    pr = torch.from_numpy(z["f32_prob"]).to(dev).requires_grad_(True)
    m2, s2 = softargmax_2d_std(pr)
    (gp,) = torch.autograd.grad([m2, s2], [pr], [ctm, cts])
    assert (m2.detach().cpu().double() - torch.from_numpy(z["f64_mean"])).abs().max().item() <= 2e-4
    assert (s2.detach().cpu().double() - torch.from_numpy(z["f64_std"])).abs().max().item() <= 2e-4
    assert rel_err(gp.cpu(), z["f32_g_prob"]) <= 2e-4


@pytest.mark.parametrize("shape", [(3, 5, 7, 9), (2, 2, 128, 128), (1, 1, 16, 20), (2, 3, 30, 50)])
def test_head_odd_shapes_vs_torch(shape):
    """Non-power-of-two / non-multiple-of-4 maps (scalar path) and the 128x128 zlmo size, vs a float64 torch restatement."""
    from lc_amd.ptnet import spatial_softargmax_2d_std
    from oracle.softargmax_oracle import spatial_softargmax_2d_std as orc

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    lg = (torch.randn(shape, generator=g) * 3)
    ctm, cts = torch.randn(shape[:2] + (2,), generator=g), torch.randn(shape[:2] + (2,), generator=g)
    x = lg.to(dev).requires_grad_(True)
    mean, std = spatial_softargmax_2d_std(x)
    (gl,) = torch.autograd.grad([mean, std], [x], [ctm.to(dev), cts.to(dev)])
    x64 = lg.double().requires_grad_(True)
    m64, s64 = orc(x64)
    (g64,) = torch.autograd.grad([m64, s64], [x64], [ctm.double(), cts.double()])
    assert (mean.detach().cpu().double() - m64.detach()).abs().max().item() <= 3e-4
    assert (std.detach().cpu().double() - s64.detach()).abs().max().item() <= 3e-4
    assert rel_err(gl.cpu(), g64) <= 3e-4


def test_head_full_size_properties():
    """BASELINE head size (256,64,64,64): probabilities sum to one => sum of the logit-gradient over each map is ~0;
    a one-hot-ish map returns its peak location; shifting the logits by a constant changes nothing."""
    from lc_amd.ptnet import spatial_softargmax_2d_std
    from lc_amd import synth

    dev = torch.device("cuda:0")
    lg = synth.make_head_logits(256, 64, 64, 64, seed=3).to(dev).requires_grad_(True)
    mean, std = spatial_softargmax_2d_std(lg)
    (g,) = torch.autograd.grad([mean.sum() + std.sum()], [lg])
    assert g.flatten(-2).sum(-1).abs().max().item() <= 1e-3
    m2, s2 = spatial_softargmax_2d_std(lg.detach() + 7.5)
    assert (m2 - mean).abs().max().item() <= 1e-4 and (s2 - std).abs().max().item() <= 1e-4
    spike = torch.full((4, 1, 64, 64), -30.0, device=dev)
    spike[:, :, 17, 42] = 30.0
    m3, s3 = spatial_softargmax_2d_std(spike)
    assert torch.allclose(m3, torch.tensor([42.0, 17.0], device=dev).expand(4, 1, 2), atol=1e-4)
    assert (s3 - 1e-3).abs().max().item() <= 1e-4  # sqrt(0 + 1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 5, 64, 64), (2, 3, 128, 128), (2, 2, 24, 40), (2, 2, 7, 9)])
def test_head_16bit_maps_match_the_fp32_kernel_on_the_same_values(dtype, shape):
    """16-bit maps are consumed natively: statistics identical to the fp32 kernel on the up-cast values, gradient = the
    fp32 gradient rounded to nearest even into the map's type (fast 64x64 / 128x128 paths, float4-able and scalar shapes)."""
    from lc_amd.ptnet import spatial_softargmax_2d_std, softargmax_2d_std

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    lg16 = (torch.randn(shape, generator=g) * 3).to(dtype).to(dev).requires_grad_(True)
    lg32 = lg16.detach().float().requires_grad_(True)
    ct_m, ct_s = torch.randn(shape[:2] + (2,), generator=g).to(dev), torch.randn(shape[:2] + (2,), generator=g).to(dev)
    m16, s16 = spatial_softargmax_2d_std(lg16)
    m32, s32 = spatial_softargmax_2d_std(lg32)
    assert m16.dtype == torch.float32 and torch.equal(m16, m32) and torch.equal(s16, s32)
    (g16,) = torch.autograd.grad((m16 * ct_m).sum() + (s16 * ct_s).sum(), lg16)
    (g32,) = torch.autograd.grad((m32 * ct_m).sum() + (s32 * ct_s).sum(), lg32)
    assert g16.dtype == dtype and torch.equal(g16, g32.to(dtype))
    # probability input (ptnet.softargmax_2d_std itself)
    p16 = lg16.detach().float().flatten(-2).softmax(-1).reshape(shape).to(dtype).requires_grad_(True)
    p32 = p16.detach().float().requires_grad_(True)
    a16, b16 = softargmax_2d_std(p16)
    a32, b32 = softargmax_2d_std(p32)
    assert torch.equal(a16, a32) and torch.equal(b16, b32)
    (h16,) = torch.autograd.grad((a16 * ct_m).sum() + (b16 * ct_s).sum(), p16)
    (h32,) = torch.autograd.grad((a32 * ct_m).sum() + (b32 * ct_s).sum(), p32)
    assert torch.equal(h16, h32.to(dtype))


def test_softargmax_1d_cov_matches_the_reference_formula():
    """ptnet.py:85-97 restated in fp64: mean = sum i p_i, cov = sum (i - mean)^2 p_i; forward and gradient."""
    from lc_amd.ptnet import softargmax_1d_cov

    g = torch.Generator().manual_seed(3)
    p = torch.rand(5, 7, 37, generator=g).softmax(-1)
    x = p.to("cuda:0").requires_grad_(True)
    m, c = softargmax_1d_cov(x)
    ct = torch.randn(5, 7, 2, generator=g)
    (gx,) = torch.autograd.grad((m * ct[..., 0].to(x.device)).sum() + (c * ct[..., 1].to(x.device)).sum(), x)
    p64 = p.double().requires_grad_(True)
    idx = torch.arange(37, dtype=torch.float64)
    m64 = (p64 * idx).sum(-1)
    c64 = (p64 * (idx - m64[..., None]) ** 2).sum(-1)
    (g64,) = torch.autograd.grad((m64 * ct[..., 0].double()).sum() + (c64 * ct[..., 1].double()).sum(), p64)
    assert m.shape == (5, 7) and (m.cpu().double() - m64).abs().max() <= 1e-5 and (c.cpu().double() - c64).abs().max() <= 1e-4
    assert (gx.cpu().double() - g64).abs().max() <= 1e-3 * g64.abs().max()


@pytest.mark.parametrize("path", golden_files("headc_"), ids=[case_name(p, "headc_") for p in golden_files("headc_")])
def test_head_vs_golden_survey_sizes(path):
    """(4,16,64,64) and (2,64,64,64) -- the sizes SURVEY.md 8c names, S = 64 being the metric's map count -- against outputs of
    the reference's ptnet.softargmax_2d_std: every map's mean/std, every map's input gradient through two random functionals,
    six maps' gradients in full (tests/golden/gen_golden.py: gen_head_compact)."""
    from lc_amd.ptnet import spatial_softargmax_2d_std
    from tests.test_oracle_head import check_compact, compact_case

    z, logits, ct_mean, ct_std, probe = compact_case(path)
    dev = torch.device("cuda:0")
    lg = logits.to(dev).requires_grad_(True)
    mean, std = spatial_softargmax_2d_std(lg)
    (g,) = torch.autograd.grad([mean, std], [lg], [ct_mean.to(dev), ct_std.to(dev)])
    check_compact(z, mean.detach(), std.detach(), g, probe, 2e-4, 2e-6, tol_map=2e-4)  # same bounds as test_head_vs_golden (fp32 kernel, __expf)
    # the fp32 reference run sits as close to the fp64 one as the kernel does
    assert (torch.from_numpy(z["f32_mean"]).double() - torch.from_numpy(z["f64_mean"])).abs().max().item() <= 2e-4
