"""The selection oracle (oracle/select_oracle.py) against goldens produced by the reference's own quantile_msk."""
import numpy as np
import pytest
import torch

from oracle import select_oracle as orc
from tests.util import golden_files, case_name

FILES = golden_files("select_")


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "select_") for p in FILES])
def test_selection_oracle_vs_reference(path):
    z = np.load(path)
    s, seg, q = torch.from_numpy(z["in_inv_std"]), torch.from_numpy(z["in_seg"]), float(z["q"])
    assert np.array_equal(orc.select_mask(s, seg, "quantile", q).numpy(), z["msk_quantile"])
    assert np.array_equal(orc.select_mask(s, seg, "quantile_in_mask", q).numpy(), z["msk_quantile_in_mask"])
    assert np.array_equal(orc.select_mask(s, seg, "mask").numpy(), z["in_seg"])
    lists = orc.select_lists(torch.from_numpy(z["msk_quantile"]))
    assert all(torch.equal(l, torch.from_numpy(np.flatnonzero(m))) for l, m in zip(lists, z["msk_quantile"]))


def test_host_quantile_msk_matches_oracle_for_float_and_per_sample_quantiles():
    from lc_amd.inference import quantile_msk

    g = torch.Generator().manual_seed(0)
    s = torch.rand(6, 97, 2, generator=g)
    assert torch.equal(quantile_msk(s, 0.3), orc.quantile_msk(s, 0.3))
    q = torch.rand(6, generator=g)
    assert torch.equal(quantile_msk(s, q), orc.quantile_msk(s, q))
