"""Pin oracle/lc_loss_oracle.py against the golden vectors produced by the unmodified reference."""
import numpy as np
import pytest
import torch

from oracle import lc_loss_oracle as orc
from tests.util import golden_files, case_name, load_loss_case, rel_err

FILES = golden_files("lc_loss_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "lc_loss_") for p in FILES])
@pytest.mark.parametrize("tag,dtype,tol_loss,tol_grad", [("f64", torch.float64, 1e-10, 1e-8), ("f32", torch.float32, 1e-4, 2e-3)])
def test_oracle_matches_reference(path, tag, dtype, tol_loss, tol_grad):
    z, ins, kwargs, want3 = load_loss_case(path, dtype)
    loss, gu, gs, gx = orc.loss_and_grads(ins["K"], ins["pose"], ins["pts3d"], ins["pts2d"], ins["inv_std"],
                                          ins.get("valid"), ins["bbox_3d"], grad_out=ins["grad_out"],
                                          want_pts3d=want3, **kwargs)
    ref_loss = torch.from_numpy(z[f"{tag}_loss"])
    # relative-to-magnitude tolerance (the z-clamp case has loss ~1e6)
    assert ((loss - ref_loss).abs() / ref_loss.abs().clamp_min(1)).max().item() <= tol_loss
    assert rel_err(gu, z[f"{tag}_g_pts2d"]) <= tol_grad
    assert rel_err(gs, z[f"{tag}_g_inv_std"]) <= tol_grad
    if want3:
        assert rel_err(gx, z[f"{tag}_g_pts3d"]) <= tol_grad


@pytest.mark.parametrize("path", [p for p in FILES if "B256" not in p and "N1024" not in p],
                         ids=lambda p: case_name(p, "lc_loss_"))
def test_oracle_intermediates_f64(path):
    z, ins, kwargs, _ = load_loss_case(path, torch.float64)
    _, inter = orc.loss_cov_mixed(ins["K"], ins["pose"], ins["pts3d"], ins["pts2d"], ins["inv_std"], ins.get("valid"),
                                  bbox_3d=ins["bbox_3d"], return_intermediates=True, **kwargs)
    for k in ("w", "c", "Hinv", "A", "G", "e"):
        if "f64_" + k in z.files:  # the largest fixtures keep loss, gradients and Hinv only (gen_golden.py)
            assert rel_err(inter[k], z["f64_" + k]) <= 1e-9, k
