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


@pytest.mark.parametrize("B,N,q", [(5, 1849, 0.7), (2, 4096, 0.9), (3, 300, 0.33), (1, 1, 0.5), (2, 5, 0.5), (3, 16384, 0.2), (2, 16384, 0.8), (2, 9000, 0.5)])
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


@pytest.mark.parametrize("mode", ["mask", "quantile", "quantile_in_mask"])
@pytest.mark.parametrize("B,H,W,sample,top_left,min_count", [(64, 64, 64, 2, (0, 0), 4), (5, 32, 32, 1, (0, 0), 4), (3, 37, 45, 2, (1, 0), 4),
                                                           (4, 64, 64, 3, (2, 1), 6), (2, 16, 16, 4, (0, 0), 12), (3, 86, 86, 2, (0, 0), 4), (2, 128, 128, 2, (0, 0), 4), (2, 96, 80, 1, (0, 0), 4),
                                                           (64, 128, 128, 1, (0, 0), 4), (3, 128, 128, 1, (0, 1), 4), (2, 181, 181, 2, (1, 1), 4)])  # zlmo test time: 16384 candidates per object
def test_front_end_and_selection_in_one_launch(B, H, W, sample, top_left, min_count, mode):
    """lc_dense_frontend_select3 (no workspace) against lc_dense_frontend_fwd3 followed by lc_dense_select_f32 on its rows and visibility
    mask: counts, source indices and every selected value bit for bit (same log-sum-exp reduction, same per-pixel arithmetic) --
    incl. an object nothing of which is visible (padded with min_count pseudo-random entries) and one that is visible everywhere."""
    from lc_amd.dense import dense_front_end_select, dense_front_end_with_visibility, dense_select

    g = torch.Generator().manual_seed(B * H + W)
    xyz = torch.randn(B, 3, H, W, generator=g).to(DEV)
    wl = (torch.randn(B, 2, H, W, generator=g) * 2).to(DEV)
    ws = (torch.rand(B, generator=g) * 50 + 1).to(DEV)
    ns = (torch.rand(B, 3, generator=g) * 100 + 10).to(DEV)
    vl = (torch.randn(B, 1, H, W, generator=g) * 3).to(DEV)
    vl[0] = -9.0
    vl[1] = 9.0
    kw = dict(quantile=0.35, square_weights=True, min_count=min_count, seed=7)
    got = dense_front_end_select(xyz, wl, ws, ns, vl, mode, seg_thresh=0.5, sample=sample, top_left=top_left, **kw)
    u, s, x, vis = dense_front_end_with_visibility(xyz, wl, ws, ns, vl, 0.5, sample=sample, top_left=top_left)
    want = dense_select(u, s, x, mode, mask=vis, **kw)
    cnt = want[3]
    assert torch.equal(got[3], cnt)
    N = u.shape[1]
    if mode != "quantile":
        assert int(cnt[0]) == (min_count if N > min_count else 0) and int(cnt[1]) >= min_count
    live = torch.arange(N, device=DEV)[None, :] < cnt[:, None]
    for k, name in ((0, "pts2d"), (1, "weights"), (2, "pts3d"), (4, "index")):
        m = live if got[k].dim() == 2 else live[..., None].expand_as(got[k])
        assert torch.equal(got[k][m], want[k][m]), name


def test_front_end_select_rejects_more_than_16384_points():
    from lc_amd.dense import dense_front_end_select

    z = torch.zeros(1, 3, 130, 128, device=DEV)
    with pytest.raises(RuntimeError, match="16384"):
        dense_front_end_select(z, z[:, :2], torch.ones(1, device=DEV), None, z[:, :1], "mask", sample=1)


@pytest.mark.parametrize("B,H,W,sample,top_left", [(64, 128, 128, 1, (0, 0)), (24, 128, 128, 1, (0, 0)), (100, 90, 90, 1, (0, 0)), (5, 200, 160, 2, (1, 0)),
                                                   (64, 96, 100, 1, (0, 0)), (3, 65, 64, 1, (0, 0))])
@pytest.mark.parametrize("mode,q", [("quantile_in_mask", 0.2), ("quantile", 0.3), ("mask", 0.0), ("quantile_in_mask", 0.999), ("quantile", 0.0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_several_workgroups_per_object_select_what_one_workgroup_selects(B, H, W, sample, top_left, mode, q, dtype):
    """`lc_dense_frontend_select3` with a workspace (rows of more than 4096 candidates: 2 / 4 / 8 workgroups per object, the shares of the
    log-sum-exp, of the radix select's histograms and of the compaction meeting through the workspace) against the one-workgroup launch: rows,
    weights, points, indices and counts bit for bit -- incl. rows that do not fill the last part, an object with nothing visible (padded), all
    weights equal (every key in one bin), repeated launches on the same workspace, maps in 16 bits."""
    from lc_amd.dense import dense_front_end_select

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * H + W)
    xyz = torch.randn(B, 3, H, W, generator=g).to(dev, dtype)
    wl = (torch.randn(B, 2, H, W, generator=g) * 1.5)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    blob = (((yy - H / 2) / (0.3 * H)) ** 2 + ((xx - W / 2) / (0.25 * W)) ** 2 < 1).float()
    wl = wl + 4 * blob
    vl = (blob * 2 - 1) * 3 + torch.randn(B, 1, H, W, generator=g)
    vl[1] = -5.0      # nothing visible: modes with the mask keep nothing, the row is padded
    wl[2] = 0.25      # all weights equal
    wl, vl = wl.to(dev, dtype), vl.to(dev, dtype)
    ws = (torch.rand(B, generator=g) + 0.5).to(dev)
    ns = (torch.rand(B, 3, generator=g) + 0.5).to(dev)
    kw = dict(seg_thresh=0.5, sample=sample, top_left=top_left, quantile=q, min_count=6, seed=3)
    one = dense_front_end_select(xyz, wl, ws, ns, vl, mode, split=False, **kw)
    N = one[0].shape[1]
    for rep in range(3):
        many = dense_front_end_select(xyz, wl, ws, ns, vl, mode, split=True, **kw)
        cnt = one[3]
        assert torch.equal(many[3], cnt), (many[3] - cnt).nonzero().flatten().tolist()
        live = torch.arange(N, device=dev)[None, :] < cnt[:, None]
        for name, x, y in zip(("pts2d", "weights", "pts3d", "", "index"), many, one):
            if name:
                m = live if x.dim() == 2 else live[..., None].expand_as(x)
                assert torch.equal(x[m], y[m]), name
    # the selection alone (binary-code heads: the points are decoded afterwards)
    one = dense_front_end_select(None, wl, ws, None, vl, mode, split=False, **kw)
    many = dense_front_end_select(None, wl, ws, None, vl, mode, split=True, **kw)
    assert torch.equal(many[3], one[3])
    live = torch.arange(N, device=dev)[None, :] < one[3][:, None]
    assert torch.equal(many[4][live], one[4][live]) and torch.equal(many[1][live[..., None].expand_as(many[1])], one[1][live[..., None].expand_as(one[1])])


def test_split_selection_soak_over_random_shapes():
    """Forty random batches (objects, map sizes, strides, offsets, modes, quantiles, element types): several workgroups per object select exactly what
    one workgroup selects -- counts, indices, rows."""
    from lc_amd import _lib
    from lc_amd.dense import dense_front_end_select

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    lib = _lib.load()
    tried = 0
    for case in range(60):
        B = int(torch.randint(1, 129, (1,), generator=g))
        H, W = (int(v) for v in torch.randint(48, 161, (2,), generator=g))
        sample = int(torch.randint(1, 3, (1,), generator=g))
        top, left = (int(v) for v in torch.randint(0, sample, (2,), generator=g))
        if lib.lc_dense_frontend_select_workspace_bytes(B, H, W, top, left, sample) == 0:
            continue
        tried += 1
        dtype = (torch.float32, torch.bfloat16, torch.float16)[case % 3]
        mode = ("quantile_in_mask", "quantile", "mask")[(case // 3) % 3]
        q = float(torch.rand(1, generator=g))
        wl = (torch.randn(B, 2, H, W, generator=g) * 2).to(dev, dtype)
        vl = (torch.randn(B, 1, H, W, generator=g) * 2 + 0.5).to(dev, dtype)
        xyz = torch.randn(B, 3, H, W, generator=g).to(dev, dtype)
        ws = (torch.rand(B, generator=g) + 0.5).to(dev)
        kw = dict(seg_thresh=0.5, sample=sample, top_left=(top, left), quantile=q, min_count=5, seed=case)
        one = dense_front_end_select(xyz, wl, ws, None, vl, mode, split=False, **kw)
        many = dense_front_end_select(xyz, wl, ws, None, vl, mode, split=True, **kw)
        assert torch.equal(many[3], one[3]), (case, B, H, W, sample, mode, q)
        live = torch.arange(one[0].shape[1], device=dev)[None, :] < one[3][:, None]
        for x, y in zip(many, one):
            if x.dim() >= 2:
                m = live if x.dim() == 2 else live[..., None].expand_as(x)
                assert torch.equal(x[m], y[m]), (case, B, H, W, sample, mode, q)
    assert tried >= 20
