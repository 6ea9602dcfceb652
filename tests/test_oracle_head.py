"""Pin oracle/softargmax_oracle.py against golden vectors produced by the reference's ptnet.softargmax_2d_std."""
import numpy as np
import pytest
import torch

from oracle import softargmax_oracle as orc
from tests.util import golden_files, case_name, rel_err

FILES = golden_files("head_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "head_") for p in FILES])
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-11), ("f32", torch.float32, 2e-5)])
def test_head_oracle(path, tag, dtype, tol):
    z = np.load(path)
    lg = torch.from_numpy(z["in_logits"]).to(dtype).requires_grad_(True)
    mean, std = orc.spatial_softargmax_2d_std(lg)
    (g,) = torch.autograd.grad([mean, std], [lg], [torch.from_numpy(z["in_ct_mean"]).to(dtype), torch.from_numpy(z["in_ct_std"]).to(dtype)])
    assert (mean.detach().double() - torch.from_numpy(z[f"{tag}_mean"]).double()).abs().max().item() <= tol * 64
    assert (std.detach().double() - torch.from_numpy(z[f"{tag}_std"]).double()).abs().max().item() <= tol * 64
    assert rel_err(g, z[f"{tag}_g_logits"]) <= tol * 10


COMPACT = golden_files("headc_")


def compact_case(path):
    """Inputs of a compact head fixture: logits from the stored int16 grid (exact on every machine), cotangents as stored, the
    probe tensors regenerated from the seed and checked bit for bit against the stored checksum."""
    from tests.golden.gen_golden import head_compact_inputs

    z = np.load(path)
    B, S = int(z["in_shape"][0]), int(z["in_shape"][1])
    _, logits, _, _, probe = head_compact_inputs(B, S, int(z["in_seed"]), logits_q=torch.from_numpy(z["in_logits_q"]))
    assert np.bitwise_xor.reduce(probe.numpy().view(np.uint64).ravel()) == z["in_probe_xor"]  # integer stream: same bits everywhere
    return z, logits, torch.from_numpy(z["in_ct_mean"]), torch.from_numpy(z["in_ct_std"]), probe


def check_compact(z, mean, std, g, probe, tol_px, tol_g, tol_map=2e-7):
    """mean/std of every map; the input gradient of every map through the two stored random functionals; six maps in full."""
    assert (mean.double().cpu() - torch.from_numpy(z["f64_mean"])).abs().max().item() <= tol_px
    assert (std.double().cpu() - torch.from_numpy(z["f64_std"])).abs().max().item() <= tol_px
    g = g.double().cpu()
    got = (g[None] * probe).sum((-1, -2))
    scale = torch.from_numpy(z["f64_g_absmax"]).double() * 64  # |sum of 4096 terms g*N(0,1)| ~ 64 x typical |g|
    assert ((got - torch.from_numpy(z["f64_g_probe"])).abs() / scale).max().item() <= tol_g
    for (b, s), ref in zip(z["g_maps"], z["f64_g_logits_maps"]):
        assert rel_err(g[b, s], ref) <= tol_map  # (the stored maps are float32 roundings of the float64 gradient: 6e-8)


@pytest.mark.parametrize("path", COMPACT, ids=[case_name(p, "headc_") for p in COMPACT])
def test_head_oracle_survey_sizes(path):
    """(4,16,64,64) and (2,64,64,64): the sizes SURVEY.md 8c asks for (S = 64 = the metric's map count)."""
    z, logits, ct_mean, ct_std, probe = compact_case(path)
    lg = logits.double().requires_grad_(True)
    mean, std = orc.spatial_softargmax_2d_std(lg)
    (g,) = torch.autograd.grad([mean, std], [lg], [ct_mean.double(), ct_std.double()])
    check_compact(z, mean.detach(), std.detach(), g, probe, 1e-9, 1e-11)
