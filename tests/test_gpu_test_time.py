"""The test-time path (`test.py:67-136`) at the reference's OWN knobs, stage by stage against the composed CPU oracles and end to end:

  zlmo (configs/zlmo.yaml:30-37): 128x128 maps, pnp_solver.dense_sample 1 -> 16 384 candidates per object, quantile_in_mask 0.2, rel_reproj_err with a
        per-object out_pix_scale, solvers [weighted_filtered], 21 binary code planes (bit_cnt 7,7,7) + model_transform;
  glmo (configs/glmo.yaml:28-32): 64x64 maps, default stride 2 -> 1024 candidates, quantile 0.3, solvers [weighted], continuous xyz head.

Every stage's oracle is fed the GPU's output of the stage before it, so each comparison is as sharp as that stage allows: decode and
front end to float32 rounding, the selection's index sets / counts / gathered values EXACTLY (select_oracle = torch.quantile, pinned by the
reference's quantile_msk incl. select_q20_B2_N16384), the RANSAC's integer outputs EXACTLY for every pose (float32-faithful oracle on the kernel's hypotheses),
the inlier re-selection exactly, the LM solves to the 1e-4 pose tolerance.  The end-to-end call must then return the stage-wise result."""
import numpy as np
import pytest
import torch

from tests.pnp_cases import pose_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to(d, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


@pytest.mark.parametrize("name,B", [("zlmo", 6), ("glmo", 8)])
def test_test_time_path_stage_by_stage(name, B):
    from lc_amd import floatbits, synth
    from lc_amd.config import AttrDict
    from lc_amd.dense import dense_front_end_select, dense_front_end_with_visibility, dense_select
    from lc_amd.inference import solve_pnp
    from lc_amd.pnp import gpu_solver, pnp_ceres
    from oracle import dense_oracle, floatbits_oracle, p3p_ransac_oracle, pnp_oracle, select_oracle

    cfg, gt_c, out_c = synth.test_time_inputs(name, B=B, seed=11)
    if name == "zlmo":
        out_c["msk_vis_logits"][1] = -9.0  # an object the network sees nothing of: nothing selected, 4 pseudo-random pad entries
    cfg = AttrDict(cfg)
    gt, out = _to(gt_c, DEV), _to(out_c, DEV)
    stride, thr_seg, mode, q = cfg.get("dense_sample", 2), cfg.get("seg_thresh", 0.5), cfg.dense_point_select, cfg.quantile
    H, W = out_c["xyz_weight_logits"].shape[-2:]
    N = -(-H // stride) * -(-W // stride)
    assert N == (16384 if name == "zlmo" else 1024)

    # ---- stage 1: network output -> object coordinates (zlmo: Gray decode x noc_scale, inverse model transform; losses.py:17-47) ----
    if name == "zlmo":
        planes = floatbits.nn_logits2xyz_planes(out["xyz_noc_bin"], gt["bit_cnt"], gt["noc_scale"], gt["model_transform"])
        noc = floatbits_oracle.nn_logits2noc(out_c["xyz_noc_bin"].double(), gt_c["bit_cnt"])
        T = gt_c["model_transform"].double()
        want = (noc * gt_c["noc_scale"].double()[:, None, None, :] - T[:, None, None, :3, 3]) @ T[:, None, :3, :3]
        assert (planes.cpu().double() - want.permute(0, 3, 1, 2)).abs().max() < 2e-4  # mm, float32 rounding of ~60 mm coordinates
        xyz_map, noc_scale = planes, None
        xyz_c, ns_c = planes.cpu(), None
    else:
        xyz_map, noc_scale = out["xyz_noc"], gt["noc_scale"]
        xyz_c, ns_c = out_c["xyz_noc"], gt_c["noc_scale"]

    # ---- stage 2: dense front end (joint softmax x scale, (0,0)-phase sub-sampling, visibility; test.py:70-92) ----
    u, s, x, vis = dense_front_end_with_visibility(xyz_map, out["xyz_weight_logits"], out["xyz_weights_scale"], noc_scale, out["msk_vis_logits"],
                                                   thr_seg, sample=stride)
    ou, os_, ox = dense_oracle.dense_front_end(xyz_c, out_c["xyz_weight_logits"], out_c["xyz_weights_scale"], ns_c, stride, (0, 0))
    assert torch.equal(u.cpu(), ou.expand(B, N, 2))
    assert ((s.cpu() - os_).abs() <= 2e-5 * os_.abs() + 1e-12).all()
    assert ((x.cpu() - ox).abs() <= 1e-6 * ox.abs() + 1e-9).all()
    prob = torch.sigmoid(out_c["msk_vis_logits"][:, 0, ::stride, ::stride]).reshape(B, N)
    sure = (prob - thr_seg).abs() > 1e-6
    assert torch.equal(vis.cpu()[sure], (prob > thr_seg)[sure])

    # ---- stage 3: point selection (test.py:94-113) -- the fused launch the pipeline takes, EXACT against torch.quantile on the GPU's own rows ----
    sel = dense_front_end_select(xyz_map, out["xyz_weight_logits"], out["xyz_weights_scale"], noc_scale, out["msk_vis_logits"], mode,
                                 seg_thresh=thr_seg, sample=stride, quantile=float(q), square_weights=True, min_count=4)
    su, sw, sx, sc, si = sel
    keep = select_oracle.select_mask(s.cpu(), vis.cpu(), mode, q)
    two = dense_select(u, s, x, mode, mask=vis, quantile=float(q), square_weights=True, min_count=4)
    for b in range(B):
        idx = keep[b].nonzero()[:, 0]
        c = int(sc[b])
        if len(idx) < 4:  # padded to min_count (test.py:108-113): the survivors first, then valid source rows
            assert c == 4 and torch.equal(si[b, :len(idx)].cpu().long(), idx) and bool(((si[b, :4] >= 0) & (si[b, :4] < N)).all())
            idx = si[b, :4].cpu().long()
        assert c == len(idx) and torch.equal(si[b, :c].cpu().long(), idx), (b, c, len(idx))
        assert torch.equal(su[b, :c].cpu(), u[b].cpu()[idx]) and torch.equal(sx[b, :c].cpu(), x[b].cpu()[idx])
        assert torch.equal(sw[b, :c].cpu(), s[b].cpu()[idx] ** 2)  # icov = inv_std^2 (test.py:92)
        for a, t in zip(sel, two):  # ... and the two launches give the same rows bit for bit
            assert torch.equal(a[b][:c] if a.dim() > 1 else a[b], t[b][:c] if t.dim() > 1 else t[b])
    counts = sc.cpu().numpy()
    if name == "zlmo":
        assert int(sc[1]) == 4 and 0.7 * 0.15 * N < np.median(counts) < 0.8 * 0.35 * N  # 80 % of the visible pixels
        assert counts.max() > 2048  # more than the prefix rounds 1-3 sampled and scored
    else:
        assert (counts == N - int(np.ceil(q * (N - 1)))).all() or (np.abs(counts - 0.7 * N) < 3).all()

    # ---- stage 4: RANSAC initialiser on the selected rows (cv2_solver.solve's place, test.py:115-120) ----
    if cfg.get("rel_reproj_err", False):
        thr = (2 / gt["out_pix_scale"]).float()
    else:
        thr = 3.0
    filt = dict(weights=sw, index=si, min_count=4)
    ws = []
    st, inl, bad, hyp, n_in = gpu_solver.solve_device(gt["out_K"], sx, su, sc, reprojectionError=thr, refine=False, return_hypothesis=True,
                                                      select=filt if "weighted_filtered" in cfg.solvers else None, workspace_out=ws)
    Kc, sxc, suc = gt_c["out_K"].numpy(), sx.cpu().numpy(), su.cpu().numpy()
    thr_c = thr.cpu().numpy() if isinstance(thr, torch.Tensor) else np.full(B, thr, np.float32)
    hyp_c, n_c, inl_c, bad_c = hyp.cpu().numpy(), n_in.cpu().numpy(), inl.cpu().numpy(), bad.cpu().numpy()
    # (a) EXACT, every pose: the float32-faithful oracle on the kernel's own hypotheses (read back from the workspace) -- winner, inlier count,
    #     inlier mask, validity (oracle/p3p_ransac_oracle.py: ransac_f32 restates the kernel's division-free float32 scoring operation by operation)
    hyp64, hyp32, _pc, _pe = (v.cpu().numpy() for v in gpu_solver.workspace_views(ws[0], B, N, 150))
    exact = [p3p_ransac_oracle.ransac_f32(Kc[b], sxc[b], suc[b], int(counts[b]), float(thr_c[b]), hyp32[b]) for b in range(B)]
    for b, r in enumerate(exact):
        assert int(bad_c[b]) == r["invalid"] and int(hyp_c[b]) == r["best_hyp"] and int(n_c[b]) == r["n_inliers"], (b, hyp_c[b], r["best_hyp"], n_c[b], r["n_inliers"])
        assert np.array_equal(inl_c[b].astype(bool), r["inlier_mask"]), b
    # (b) the sanity bound: the independent float64 P3P + float64 scoring of the same oracle.  Where it says float32 has no freedom it names the
    #     same winner; its pose is the kernel's to 1e-4; its count differs by no more than the points it sees within 1e-3 of the threshold
    res = [p3p_ransac_oracle.ransac(Kc[b], sxc[b], suc[b], int(counts[b]), float(thr_c[b]), 150, 0, b) for b in range(B)]
    decided = 0
    for b, r in enumerate(res):
        assert int(bad_c[b]) == r["invalid"], b
        if r["invalid"]:
            continue
        if r["decided"]:
            decided += 1
            assert hyp_c[b] == r["best_hyp"], (b, hyp_c[b], r["best_hyp"])
            dq, dt = pose_err(st[b:b + 1].cpu().numpy(), np.concatenate((p3p_ransac_oracle.rot_to_quat(r["R"]), r["t"]))[None])
            assert dq.max() <= 1e-4 and dt.max() <= 1e-4
            open_pts, want_in = np.zeros(N, bool), np.zeros(N, bool)
            open_pts[r["mask_unsure"]], want_in[r["inliers"]] = True, True
            assert (inl_c[b] == want_in)[~open_pts].all() and abs(int(n_c[b]) - r["n_inliers"]) <= open_pts.sum() <= 0.01 * counts[b] + 2
        else:
            assert r["per_hyp_count"][hyp_c[b]] >= r["per_hyp_count"].max() - 2
    live = [b for b in range(B) if not (name == "zlmo" and b == 1)]
    assert (n_c[live] > 0.4 * counts[live]).all() if name == "zlmo" else (n_c[live] > 30).all()

    # ---- stage 5: 'weighted-filtered' re-selection = the selection intersected with the inliers (test.py:129-131) ----
    if "weighted_filtered" in cfg.solvers:
        fu, fw, fx, fc, fi = filt["result"]
        for b in range(B):
            pos = np.flatnonzero(inl_c[b][:counts[b]])
            c = int(fc[b])
            if len(pos) < 4:
                assert c == (4 if counts[b] > 4 else len(pos))
                continue
            assert c == len(pos)
            assert torch.equal(fi[b, :c].cpu(), si[b].cpu()[pos]) and torch.equal(fu[b, :c].cpu(), su[b].cpu()[pos])
            assert torch.equal(fw[b, :c].cpu(), sw[b].cpu()[pos]) and torch.equal(fx[b, :c].cpu(), sx[b].cpu()[pos])

    # ---- stage 6: the two LM solves (the RANSAC's inlier refinement, then the weighted solve from its result; cer_solver.solve, test.py:122-134) ----
    rows = torch.where(bad, torch.zeros_like(sc), sc)
    ref_state, _, _ = pnp_ceres.solve_device(gt["out_K"], sx, su, None, st, rows, max_iter_count=20, weight_mask=inl)
    unit = np.zeros((B, N, 2, 2), np.float32)
    unit[..., 0, 0] = unit[..., 1, 1] = inl_c
    o_ref, _, _ = pnp_oracle.solve_batched(st.cpu().numpy(), Kc, suc, sxc, unit, rows.cpu().numpy(), max_iter=20)
    dq, dt = pose_err(ref_state.cpu().numpy()[live], o_ref[live])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, ("refinement", dq, dt)
    if "weighted_filtered" in cfg.solvers:
        wu, ww, wx, wc = fu, fw, fx, fc
        key = "weighted-filtered"
    else:
        wu, ww, wx, wc = su, sw, sx, sc
        key = "weighted"
    w_state, _, w_ret = pnp_ceres.solve_device(gt["out_K"], wx, wu, ww, ref_state, wc, weights_are_icov=True, nan_to_num=True)
    L = np.zeros((B, N, 2, 2), np.float32)
    L[..., 0, 0], L[..., 1, 1] = np.sqrt(np.abs(ww.cpu().numpy()[..., 0])), np.sqrt(np.abs(ww.cpu().numpy()[..., 1]))  # (entries behind a row's count are not defined)
    o_w, _, o_ret = pnp_oracle.solve_batched(ref_state.cpu().numpy(), Kc, wu.cpu().numpy(), wx.cpu().numpy(), L, wc.cpu().numpy())
    assert np.array_equal(w_ret.cpu().numpy()[live], o_ret[live])
    dq, dt = pose_err(w_state.cpu().numpy()[live], o_w[live])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (key, dq, dt)

    # ---- end to end: the pipeline (fused selection, RANSAC with the re-selection inside, chained solves) returns the stage-wise result ----
    got = solve_pnp(cfg, out, gt)
    assert list(got) == [key]
    assert torch.equal(got[key], w_state), (got[key] - w_state).abs().max()
    dq, dt = pose_err(got[key].cpu().numpy()[live], gt_c["pose_best"].numpy()[live])
    print(f"{name}: selected {counts.tolist()}, inliers {n_c.tolist()}, decided {decided}/{len(live)}, pose error vs ground truth dq {dq.max():.2e} dt {dt.max():.2e}")
    assert dq.max() < 5e-2 and dt.max() < 2e-2  # 0.4-1 mm of coordinate noise on a 40 mm object


@pytest.mark.parametrize("name", ["zlmo", "glmo"])
def test_test_time_path_batch_of_64_and_graph_replay(name):
    """The benchmark's shape (64 objects): poses close to the ground truth, and the captured pipeline replays to the eager result on new inputs."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import GraphedSolvePnP, solve_pnp

    cfg, gt_c, out_c = synth.test_time_inputs(name, B=64, seed=5)
    cfg = AttrDict(cfg)
    gt, out = _to(gt_c, DEV), _to(out_c, DEV)
    key = "weighted-filtered" if name == "zlmo" else "weighted"
    ref = solve_pnp(cfg, out, gt)
    dq, dt = pose_err(ref[key].cpu().numpy(), gt_c["pose_best"].numpy())
    assert np.median(dq) < 1e-2 and dq.max() < 5e-2 and dt.max() < 2e-2, (dq.max(), dt.max())
    solver = GraphedSolvePnP(cfg, out, gt)
    _, gt2, out2 = synth.test_time_inputs(name, B=64, seed=6)
    gt2, out2 = _to(gt2, DEV), _to(out2, DEV)
    assert torch.equal(solver(out2, gt2)[key], solve_pnp(cfg, out2, gt2)[key])
    assert torch.equal(solver(out, gt)[key], ref[key])


@pytest.mark.parametrize("parts", [2, 4, 5])
def test_sub_batches_on_streams_return_the_one_batch_result(parts, monkeypatch):
    """zlmo's chain cut into sub-batches that run on side streams (what `solve_pnp_dense` does at 16 384 candidates per object): every pose
    bit for bit the pose of the one call over the whole batch -- the hypothesis streams and padding draws are keyed by the object's index in
    the whole batch -- incl. an object nothing of which is visible (padded entries) and a split that does not divide the batch."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp

    cfg, gt_c, out_c = synth.test_time_inputs("zlmo", B=37, seed=8)
    out_c["msk_vis_logits"][20] = -9.0
    cfg = AttrDict(cfg)
    gt, out = _to(gt_c, DEV), _to(out_c, DEV)
    monkeypatch.setenv("LC_AMD_TEST_TIME_STREAMS", "1")
    one = solve_pnp(cfg, out, gt)
    monkeypatch.setenv("LC_AMD_TEST_TIME_STREAMS", str(parts))
    cut = solve_pnp(cfg, out, gt)
    torch.cuda.synchronize()
    assert list(one) == list(cut) and all(torch.equal(one[k], cut[k]) for k in one)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,bits,stride", [(5, 128, 128, (7, 7, 7), 1), (3, 37, 45, (5, 7, 3), 2), (2, 64, 64, (6, 6, 5), 1)])
def test_decode_of_the_selected_rows_equals_the_whole_map_decode(B, H, W, bits, stride, dtype):
    """Binary-code heads at test time: the selection without model points followed by `decode_selected_rows` (codes of the selected pixels only) leaves
    the rows that the whole-map decode + selection leaves, bit for bit -- padding entries of an invisible object included."""
    from lc_amd import floatbits
    from lc_amd.dense import dense_front_end_select

    g = torch.Generator().manual_seed(H + W)
    C = sum(bits)
    logits = (torch.randn(B, C, H, W, generator=g) * 2).to(DEV).to(dtype)
    wl = (torch.randn(B, 2, H, W, generator=g) * 2).to(DEV).to(dtype)
    vl = (torch.randn(B, 1, H, W, generator=g) * 3).to(DEV).to(dtype)
    vl[0] = -9.0
    ws = (torch.rand(B, generator=g) * 30 + 5).to(DEV)
    ns = (torch.rand(B, 3, generator=g) * 50 + 20).to(DEV)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, :3] = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    T[:, :3, 3] = torch.randn(B, 3, generator=g)
    T = T.to(DEV)
    kw = dict(seg_thresh=0.5, sample=stride, quantile=0.2, min_count=4, seed=3)
    planes = floatbits.nn_logits2xyz_planes(logits, list(bits), ns, T)
    want = dense_front_end_select(planes, wl, ws, None, vl, "quantile_in_mask", **kw)
    got = dense_front_end_select(None, wl, ws, None, vl, "quantile_in_mask", **kw)
    got[2].fill_(float("nan"))  # the selection alone does not touch the model points
    floatbits.decode_selected_rows(logits, list(bits), got[4], got[3], got[2], noc_scale=ns, model_transform=T, sample=stride)
    cnt = want[3]
    assert torch.equal(got[3], cnt) and int(cnt[0]) == 4
    live = torch.arange(got[0].shape[1], device=DEV)[None, :] < cnt[:, None]
    for k in (0, 1, 2, 4):
        m = live if got[k].dim() == 2 else live[..., None].expand_as(got[k])
        assert torch.equal(got[k][m], want[k][m]), k
    assert torch.isnan(got[2][~live]).all()  # nothing behind a row's count is written
