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
