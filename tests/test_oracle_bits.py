"""Pin oracle/floatbits_oracle.py and lc_amd.floatbits.nn_noc2target against goldens from the reference's floatbits.py."""
import numpy as np
import pytest
import torch

from oracle import floatbits_oracle as orc
from tests.util import golden_files, case_name, rel_err

FILES = golden_files("bits_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "bits_") for p in FILES])
def test_bits_oracle(path):
    from lc_amd import floatbits as fb

    z = np.load(path)
    bits = [int(b) for b in z["bits"]]
    mod, raw = fb.nn_noc2target(torch.from_numpy(z["in_noc"]), bits)
    assert np.array_equal(mod.numpy(), z["in_mod_bits"]) and np.array_equal(raw.numpy(), z["in_raw_bits"])
    for tag, dt, tol in (("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-6)):
        lg = torch.from_numpy(z["in_logits"]).to(dt).requires_grad_(True)
        out = orc.nn_logits2noc_with_gt(lg, torch.from_numpy(z["in_raw_bits"]), bits, torch.from_numpy(z["in_msk"]))
        (gl,) = torch.autograd.grad(out, lg, torch.from_numpy(z["in_ct"]).to(dt))
        assert rel_err(out.detach(), z[f"{tag}_noc_gt"]) <= tol and rel_err(gl, z[f"{tag}_g_logits"]) <= tol
        inf = orc.nn_logits2noc(torch.from_numpy(z["in_logits"]).to(dt), bits)
        assert rel_err(inf, z[f"{tag}_noc_inf"]) <= tol
