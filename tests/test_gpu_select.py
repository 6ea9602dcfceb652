"""GPU parity of the dense point selection + compaction kernel (test-time half) -- bit-exact index sets."""
import numpy as np
import pytest
import torch

from tests.util import golden_files, case_name

pytestmark = pytest.mark.gpu
FILES = golden_files("select_")
DEV = "cuda:0"


def _inputs(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, N, 2, generator=g) * 64, torch.rand(B, N, 2, generator=g) * 2 + 0.01, torch.randn(B, N, 3, generator=g),
            torch.rand(B, N, generator=g) > 0.4)


def _check_rows(out, src_u, src_s, src_x, expect_mask, square):
    o_u, o_w, o_x, o_c, o_i = (t.cpu() for t in out)
    for b in range(src_u.shape[0]):
        idx = torch.from_numpy(np.flatnonzero(expect_mask[b]))
        c = int(o_c[b])
        assert c == len(idx), (b, c, len(idx))
        assert torch.equal(o_i[b, :c].long(), idx)
        assert torch.equal(o_u[b, :c], src_u[b, idx]) and torch.equal(o_x[b, :c], src_x[b, idx])
        assert torch.equal(o_w[b, :c], src_s[b, idx] ** 2 if square else src_s[b, idx])


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "select_") for p in FILES])
def test_select_vs_reference_golden(path):
    from lc_amd.dense import dense_select

    z = np.load(path)
    s, seg, q = torch.from_numpy(z["in_inv_std"]), torch.from_numpy(z["in_seg"]), float(z["q"])
    B, N = seg.shape
    u, _, x, _ = _inputs(B, N, 1)
    args = (u.to(DEV), s.to(DEV), x.to(DEV))
    _check_rows(dense_select(*args, "quantile", quantile=q, min_count=0), u, s, x, z["msk_quantile"], True)
    _check_rows(dense_select(*args, "quantile_in_mask", mask=seg.to(DEV), quantile=q, min_count=0, square_weights=False), u, s, x,
                z["msk_quantile_in_mask"], False)
    _check_rows(dense_select(*args, "mask", mask=seg.to(DEV), min_count=0), u, s, x, z["in_seg"], True)


@pytest.mark.parametrize("B,N,q", [(5, 1849, 0.7), (2, 4096, 0.9), (3, 300, 0.33), (1, 1, 0.5), (2, 5, 0.5)])
def test_select_vs_oracle_sizes(B, N, q):
    from lc_amd.dense import dense_select
    from oracle import select_oracle as orc

    u, s, x, seg = _inputs(B, N, N)
    args = (u.to(DEV), s.to(DEV), x.to(DEV))
    for mode in ("quantile", "quantile_in_mask", "mask"):
        exp = orc.select_mask(s, seg, mode, q).numpy()
        _check_rows(dense_select(*args, mode, mask=seg.to(DEV), quantile=q, min_count=0), u, s, x, exp, True)


def test_second_stage_selection_and_padding():
    """Selection of an already compacted list by an inlier mask (test.py:129-131) keeps source indices; fewer than
    min_count survivors are padded with valid source rows (test.py:108-113)."""
    from lc_amd.dense import dense_select

    B, N = 3, 200
    u, s, x, seg = _inputs(B, N, 3)
    seg[2] = False
    seg[2, 17] = True  # one survivor -> padded to 4
    o_u, o_w, o_x, o_c, o_i = dense_select(u.to(DEV), s.to(DEV), x.to(DEV), "mask", mask=seg.to(DEV), square_weights=False)
    assert o_c.tolist()[:2] == seg[:2].sum(-1).tolist() and int(o_c[2]) == 4
    i2 = o_i[2, :4].cpu().long()
    assert int(i2[0]) == 17 and bool(((i2 >= 0) & (i2 < N)).all())
    assert torch.equal(o_u[2, :4].cpu(), u[2, i2]) and torch.equal(o_x[2, :4].cpu(), x[2, i2]) and torch.equal(o_w[2, :4].cpu(), s[2, i2])
    # second stage on sample 0/1: keep every third survivor
    inl = torch.zeros(B, N, dtype=torch.bool)
    inl[:, ::3] = True
    p_u, p_w, p_x, p_c, p_i = dense_select(o_u, o_w, o_x, "mask", mask=inl.to(DEV), counts=o_c, index=o_i, square_weights=True, min_count=0)
    for b in range(2):
        src = torch.from_numpy(np.flatnonzero(seg[b].numpy()))[::3]
        c = int(p_c[b])
        assert c == len(src) and torch.equal(p_i[b, :c].cpu().long(), src)
        assert torch.equal(p_u[b, :c].cpu(), u[b, src]) and torch.equal(p_w[b, :c].cpu(), s[b, src] ** 2)


def test_select_rejects_bad_arguments():
    from lc_amd.dense import dense_select

    u, s, x, seg = _inputs(2, 8, 0)
    with pytest.raises(RuntimeError):
        dense_select(u.to(DEV), s.to(DEV), x.to(DEV), "quantile_in_mask", quantile=0.5)  # needs a mask
    with pytest.raises(RuntimeError):
        dense_select(u.to(DEV), s.to(DEV), x.to(DEV), "quantile", quantile=1.5)
