"""The GPU PnP initialiser's integer outputs with EQUALITY for every pose.

The kernel scores in division-free IEEE float32 (lc_pnp_init.hip: inlier_q / chunk_error), which `oracle/p3p_ransac_oracle.py` restates
operation by operation (`fma32`: an exactly rounded float32 fma; the error sums in the kernel's association).  Fed the kernel's own float32
hypotheses (read back from the split form's workspace), the oracle must reproduce per hypothesis the inlier count and the BITS of the
inlier error, and per pose the winner, the inlier count, the inlier mask and the validity flag -- for both launch forms, which share the
hypothesis stream.  The hypotheses themselves are held against the independent float64 P3P of the same oracle (two different algorithms
on the same minimal sample: the sanity bound; `tests/test_gpu_pnp_init_oracle.py` keeps the float64 run of the whole RANSAC).

Second half: the sparse head's test-time chain (`test.py:47-64` at `configs/gsplmo.yaml:30-34`: 16 keypoints, solvers ransac + weighted)
stage by stage under those oracles, with NaN / inf standard deviations in the rows."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import p3p_ransac_oracle as O
from tests.pnp_cases import pose_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "ransac_*.npz")))
DEV = "cuda:0"


def _run_both_forms(K, X, U, counts, thr, iterations, seed, **kw):
    from lc_amd.pnp import gpu_solver

    ws = []
    split = gpu_solver.solve_device(K, X, U, counts, reprojectionError=thr, iterations=iterations, seed=seed, refine=False, return_hypothesis=True,
                                    split=True, workspace_out=ws, **kw)
    single = gpu_solver.solve_device(K, X, U, counts, reprojectionError=thr, iterations=iterations, seed=seed, refine=False, return_hypothesis=True,
                                     split=False, **kw)
    B, N = X.shape[:2]
    torch.cuda.synchronize()
    views = [v.cpu().numpy() for v in gpu_solver.workspace_views(ws[0], B, N, iterations)]
    return split, single, views


def _check_exact(K, X, U, counts, thr_px, outs, views, what):
    """Every pose of a batch against ransac_f32 on the kernel's hypotheses; -> the oracle's results."""
    st, inl, bad, hyp, n_in = (t.cpu().numpy() for t in outs)
    hyp64, hyp32, part_cnt, part_err = views
    B, N = X.shape[:2]
    res = []
    for b in range(B):
        n = int(counts[b])
        r = O.ransac_f32(K[b], X[b], U[b], n, float(thr_px[b]), hyp32[b])
        res.append(r)
        assert int(bad[b]) == r["invalid"], (what, b)
        assert int(hyp[b]) == r["best_hyp"], (what, b, int(hyp[b]), r["best_hyp"])
        assert int(n_in[b]) == r["n_inliers"], (what, b, int(n_in[b]), r["n_inliers"])
        assert np.array_equal(inl[b].astype(bool), r["inlier_mask"]), (what, b, int((inl[b].astype(bool) != r["inlier_mask"]).sum()))
        if r["invalid"]:
            continue
        assert r["n_inliers"] == int(r["per_hyp_count"][r["best_hyp"]]), "the mask's count is the winner's count"
        # the winner's pose is its double-precision hypothesis as a quaternion
        want = np.concatenate((O.rot_to_quat(hyp64[b, r["best_hyp"], :9].reshape(3, 3)), hyp64[b, r["best_hyp"], 9:]))
        dq, dt = pose_err(st[b:b + 1], want[None])
        assert dq.max() <= 2e-6 and dt.max() <= 1e-6, (what, b, dq, dt)
    return res


def _check_partials(counts, views, res, what):
    """The split form's per-chunk partials, summed as the selection sums them: the oracle's counts exactly and error sums BIT FOR BIT."""
    _, _, part_cnt, part_err = views
    for b, r in enumerate(res):
        if r["per_hyp_count"] is None:
            continue
        C = (int(counts[b]) + 63) // 64
        cnt = part_cnt[b, :C].astype(np.int64).sum(0)
        err = np.zeros(part_err.shape[-1], np.float32)
        for c in range(C):
            err = (err + part_err[b, c]).astype(np.float32)
        assert np.array_equal(cnt, r["per_hyp_count"]), (what, b, np.flatnonzero(cnt != r["per_hyp_count"])[:5])
        assert np.array_equal(err.view(np.uint32), r["per_hyp_err"].view(np.uint32)), (what, b, np.flatnonzero(err != r["per_hyp_err"])[:5])


def _check_hypotheses_against_float64_p3p(K, X, U, counts, thr_px, iterations, seed, views, res, pose0=0, max_poses=4):
    """Sanity bound on what the exact comparison takes as given: the winning hypothesis against the independent float64 P3P on the same sample."""
    hyp64 = views[0]
    checked = 0
    for b, r in enumerate(res):
        if r["invalid"] or checked >= max_poses:
            continue
        n = int(counts[b])
        un, _ = O.normalised_points_f32(K[b], U[b, :n].astype(np.float32))
        un = un.astype(np.float64)
        idx = O.sample_indices(seed, pose0 + b, r["best_hyp"], n)
        yb = np.concatenate((un[idx[:3]], np.ones((3, 1))), 1)
        yb /= np.linalg.norm(yb, axis=1, keepdims=True)
        sols = O.p3p_grunert(yb, X[b, idx[:3]].astype(np.float64))
        Rk, tk = hyp64[b, r["best_hyp"], :9].reshape(3, 3), hyp64[b, r["best_hyp"], 9:]
        gap = min((np.abs(R - Rk).max() + np.abs(t - tk).max() / max(1.0, np.abs(tk).max()) for R, t in sols), default=np.inf)
        assert gap < 1e-6, (b, gap)
        checked += 1
    return checked


def _check_all_hypotheses_against_float64_p3p(K, X, U, counts, seed, views, poses):
    """EVERY hypothesis of the given poses, not only the winner: a pose the kernel formed is one of the independent float64 P3P's solutions of
    the same three correspondences (within 1e-6), and where the kernel formed none the float64 P3P has no solution in front of the camera for
    the fourth point either.  -> (hypotheses with a pose, of which matched, hypotheses without a pose, of which confirmed)."""
    hyp64 = views[0]
    have = matched = none = confirmed = 0
    for b in poses:
        n = int(counts[b])
        un, _ = O.normalised_points_f32(K[b], U[b, :n].astype(np.float32))
        un = un.astype(np.float64)
        Xb = X[b].astype(np.float64)
        for h in range(hyp64.shape[1]):
            idx = O.sample_indices(seed, b, h, n)
            yb = np.concatenate((un[idx[:3]], np.ones((3, 1))), 1)
            yb /= np.linalg.norm(yb, axis=1, keepdims=True)
            sols = O.p3p_grunert(yb, Xb[idx[:3]])
            Rk, tk = hyp64[b, h, :9].reshape(3, 3), hyp64[b, h, 9:]
            if not Rk.any():  # the kernel's "no usable solution" marker (R = 0, t = (0, 0, -1))
                none += 1
                confirmed += not any((R @ Xb[idx[3]] + t)[2] > 0 for R, t in sols)
            else:
                have += 1
                gap = min((np.abs(R - Rk).max() + np.abs(t - tk).max() / max(1.0, np.abs(tk).max()) for R, t in sols), default=np.inf)
                matched += gap < 1e-6
    return have, matched, none, confirmed


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[7:-4] for p in FILES])
def test_ransac_integers_equal_the_float32_oracle_on_the_fixtures(path):
    z = np.load(path)
    K, X, U, counts = z["in_K"], z["in_pts3d"], z["in_pts2d"], z["in_counts"]
    thr, iters, seed = float(z["in_reproj_err"]), int(z["in_iterations"]), int(z["in_seed"])
    dev = torch.device(DEV)
    split, single, views = _run_both_forms(*(torch.from_numpy(a).to(dev) for a in (K, X, U, counts)), thr, iters, seed)
    thr_px = np.full(len(K), thr, np.float32)
    res = _check_exact(K, X, U, counts, thr_px, split, views, "split")
    _check_exact(K, X, U, counts, thr_px, single, views, "single launch")
    _check_partials(counts, views, res, os.path.basename(path))
    assert _check_hypotheses_against_float64_p3p(K, X, U, counts, thr_px, iters, seed, views, res) >= 1
    # what the exact comparison takes as given -- the hypotheses -- held against the independent float64 P3P one by one (two poses per fixture)
    poses = [b for b in range(len(K)) if counts[b] >= 4][:2]
    have, matched, none, confirmed = _check_all_hypotheses_against_float64_p3p(K, X, U, counts, seed, views, poses)
    print(f"{os.path.basename(path)}: {matched} of {have} kernel hypotheses are float64 P3P solutions of their sample (1e-6); {confirmed} of {none} 'no solution' confirmed")
    assert have > 100 and matched >= 0.995 * have and confirmed >= 0.9 * none


@pytest.mark.parametrize("B,N,noise,outl,iters", [(6, 16, 0.5, 0.1, 150), (12, 16, 0.0, 0.0, 150), (4, 3300, 0.4, 0.25, 150), (3, 16384, 0.3, 0.3, 150),
                                                  (130, 4100, 0.5, 0.3, 64), (5, 1024, 0.0, 0.0, 192)])
def test_ransac_integers_equal_the_float32_oracle_on_seeded_batches(B, N, noise, outl, iters):
    """Shapes of the three test-time chains (16 keypoints; ~3300 selected pixels; whole 16 384-candidate rows) and of every scoring kernel
    (one-chunk, wide groups, live units), noise-free rows included -- there every hypothesis has every inlier and the ERROR decides: the
    winner is exact only if the error sums are."""
    from lc_amd import synth

    b = synth.make_batch(B, N, seed=B + N, outlier_frac=outl, noise_px=noise)
    g = torch.Generator().manual_seed(N)
    counts = torch.randint(max(4, N // 2), N + 1, (B,), generator=g).to(torch.int32)
    counts[0] = min(N, 3)  # too few -> invalid
    thr_t = (torch.rand(B, generator=g) * 2 + 1)  # a threshold per pose (rel_reproj_err)
    dev = torch.device(DEV)
    split, single, views = _run_both_forms(b["K"].to(dev), b["pts3d"].to(dev), b["pts2d"].to(dev), counts.to(dev), thr_t.to(dev), iters, 11)
    K, X, U = b["K"].numpy(), b["pts3d"].numpy(), b["pts2d"].numpy()
    cn = counts.numpy()
    if B > 16:  # the oracle walks every (hypothesis, point) pair in float64 emulation: a sample of a large batch
        rows = np.r_[0:4, B // 2:B // 2 + 4, B - 4:B]
        take = lambda t: t[rows]  # noqa: E731
        split, single = [take(t) for t in split], [take(t) for t in single]
        views = [v[rows] for v in views]
        K, X, U, cn, thr_t = K[rows], X[rows], U[rows], cn[rows], thr_t[rows]
    res = _check_exact(K, X, U, cn, thr_t.numpy(), split, views, "split")
    _check_exact(K, X, U, cn, thr_t.numpy(), single, views, "single launch")
    _check_partials(cn, views, res, (B, N))
    assert sum(1 for r in res if not r["invalid"]) >= len(res) - 1


def test_sparse_test_time_chain_stage_by_stage():
    """`test.py:47-64` at `configs/gsplmo.yaml:30-34` (16 keypoints, solvers [ransac, weighted], reprojection error 2 px), 64 objects, with NaN and
    inf standard deviations and a gross outlier share: (1) RANSAC integers and masks exact against the float32 oracle; (2) the inlier
    refinement (20 LM iterations, unit information on the inliers) within 1e-4 of `pnp_oracle` from the RANSAC pose; (3) the weighted solve on
    all keypoints with `1 / std**2` -- NaN filtered as `cer_solver.py:29-31` does -- within 1e-4 of `pnp_oracle` from the refined pose, same
    validity flags; (4) `solve_pnp` returns exactly the stage-wise result."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp
    from lc_amd.pnp import gpu_solver, pnp_ceres
    from oracle import pnp_oracle

    B, N = 64, 16
    dev = torch.device(DEV)
    b = synth.make_batch(B, N, seed=9, noise_px=0.5, outlier_frac=0.08)
    std = 1 / b["inv_std"]
    std[3, 5, 0] = float("nan")
    std[4, 2, 1] = float("inf")
    std[7, :, :] = float("nan")  # a whole object without usable deviations: every weight is filtered to 0 (kernel and oracle must agree on what that solve returns)
    out = dict(pts2d=b["pts2d"].to(dev), pts2d_std=std.to(dev))
    gt = dict(out_K=b["K"].to(dev), pts3d=b["pts3d"].to(dev))
    K, X, U = b["K"].numpy(), b["pts3d"].numpy(), b["pts2d"].numpy()
    counts = np.full(B, N, np.int32)

    # ---- stage 1: RANSAC (cv2_solver.solve's place, test.py:59) ----
    split, single, views = _run_both_forms(gt["out_K"], gt["pts3d"], out["pts2d"], None, 2.0, 150, 0)
    res = _check_exact(K, X, U, counts, np.full(B, 2.0, np.float32), single, views, "single launch (the chain's form at 16 keypoints)")
    _check_exact(K, X, U, counts, np.full(B, 2.0, np.float32), split, views, "split")
    _check_partials(counts, views, res, "gsplmo")
    assert _check_hypotheses_against_float64_p3p(K, X, U, counts, None, 150, 0, views, res, max_poses=8) == 8
    st, inl, bad, _hyp, _n = single
    assert not bool(bad.any())

    # ---- stage 2: inlier refinement ----
    rows = torch.full((B,), N, dtype=torch.int32, device=dev)
    ref_state, _, ref_ret = pnp_ceres.solve_device(gt["out_K"], gt["pts3d"], out["pts2d"], None, st, rows, max_iter_count=20, weight_mask=inl.to(torch.uint8))
    unit = np.zeros((B, N, 2, 2), np.float32)
    unit[..., 0, 0] = unit[..., 1, 1] = inl.cpu().numpy()
    o_ref, _, o_ret = pnp_oracle.solve_batched(st.cpu().numpy(), K, U, X, unit, counts, max_iter=20)
    assert np.array_equal(ref_ret.cpu().numpy(), o_ret)
    dq, dt = pose_err(ref_state.cpu().numpy(), o_ref)
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, ("refinement", dq.max(), dt.max())

    # ---- stage 3: weighted solve on all keypoints, icov = 1 / std^2 (test.py:52,62; cer_solver.py:29-36) ----
    w_state, _, w_ret = pnp_ceres.solve_device(gt["out_K"], gt["pts3d"], out["pts2d"], out["pts2d_std"], ref_state, None, weights_are_std=True, nan_to_num=True)
    icov = torch.nan_to_num(1 / (std * std))  # torch's own float operations, then the filter
    L = torch.diag_embed(icov.sqrt()).numpy()
    o_w, _, o_wret = pnp_oracle.solve_batched(ref_state.cpu().numpy(), K, U, X, L, counts)
    assert np.array_equal(w_ret.cpu().numpy(), o_wret)
    ok = o_wret == 0
    assert ok.sum() >= B - 2
    dq, dt = pose_err(w_state.cpu().numpy()[ok], o_w[ok])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, ("weighted", dq.max(), dt.max())

    # ---- end to end ----
    got = solve_pnp(AttrDict(solvers=["ransac", "weighted"]), out, gt)
    assert torch.equal(got["ransac"], ref_state) and torch.equal(got["weighted"], w_state)
    dq, dt = pose_err(got["weighted"].cpu().numpy()[ok], b["pose"].numpy()[ok])
    print(f"gsplmo chain vs ground truth: median dq {np.median(dq):.3e} dt {np.median(dt):.3e}")
    assert np.median(dq) < 0.2 and np.median(dt) < 0.1  # 16 keypoints, 0.5 px noise, 8 % gross outliers inside a NON-robust weighted solve: a sanity bound only, the parity is above
