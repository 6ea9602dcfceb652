"""The GPU PnP initialiser (lc_pnp_ransac_kernel, through the C ABI) against oracle/p3p_ransac_oracle.py on the committed fixtures:
same hypothesis stream (integer hash, restated bit for bit), an independent float64 P3P.  Integer outputs -- the winning
hypothesis index, the inlier count, the inlier index set -- are compared EXACTLY wherever the oracle says the float32 kernel has
no legitimate freedom (`decided`: no point within 1e-3 of the inlier threshold among the hypotheses in contention); everywhere
else the kernel's winner must still be among the oracle's best.  (OpenCV parity is unpinned: no OpenCV in the image.)"""
import glob
import os

import numpy as np
import pytest
import torch

from tests.pnp_cases import pose_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "ransac_*.npz")))


@pytest.mark.parametrize("split", [True, False], ids=["split", "single_launch"])
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[7:-4] for p in FILES])
def test_ransac_kernel_vs_oracle_fixture(path, split):
    from lc_amd.pnp import gpu_solver

    z = np.load(path)
    dev = torch.device("cuda:0")
    K, X, U = (torch.from_numpy(z["in_" + k]).to(dev) for k in ("K", "pts3d", "pts2d"))
    counts = torch.from_numpy(z["in_counts"]).to(dev)
    st, inl, bad, hyp, n_in = gpu_solver.solve_device(K, X, U, counts, reprojectionError=float(z["in_reproj_err"]), iterations=int(z["in_iterations"]),
                                                      seed=int(z["in_seed"]), refine=False, return_hypothesis=True, split=split)
    st, inl, bad, hyp, n_in = st.cpu().numpy(), inl.cpu().numpy(), bad.cpu().numpy().astype(np.int32), hyp.cpu().numpy(), n_in.cpu().numpy()
    np.testing.assert_array_equal(bad, z["invalid"])
    valid = z["invalid"] == 0
    assert (hyp[~valid] == -1).all() and (n_in[~valid] == 0).all() and not inl[~valid].any()
    decided = z["decided"] & valid
    name = os.path.basename(path)
    print(f"{name}: {int(decided.sum())}/{int(valid.sum())} poses decided; best hypothesis equal on {int((hyp == z['best_hyp'])[valid].sum())}")
    np.testing.assert_array_equal(hyp[decided], z["best_hyp"][decided])  # integer output, exact
    md = z["mask_decided"] & valid
    np.testing.assert_array_equal(n_in[md], z["n_inliers"][md])
    np.testing.assert_array_equal(inl[md], z["inlier_mask"][md].astype(bool))
    dq, dt = pose_err(st[decided], z["states"][decided]) if decided.any() else (np.zeros(1), np.zeros(1))
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (dq.max(), dt.max())  # two different P3P algorithms, same minimal sample
    # poses the oracle leaves open (ties / points on the threshold): the kernel's winner is still one of the oracle's best
    cnt = z["per_hyp_count"].astype(np.int32)
    best = cnt.max(1)
    rows = np.nonzero(valid)[0]
    assert (cnt[rows, hyp[rows]] >= best[rows] - 2).all(), (cnt[rows, hyp[rows]], best[rows])
    if "clean" in name:  # noise-free: every point is an inlier of the winner, and the pose is the ground truth
        assert inl[valid].all() and (n_in[valid] == z["in_counts"][valid]).all()
        dq, dt = pose_err(st[valid], z["in_pose_gt"][valid])
        assert dq.max() < 2e-4 and dt.max() < 2e-4
    else:
        assert decided.sum() >= 0.6 * valid.sum()


@pytest.mark.parametrize("noise", [0.7, 0.0], ids=["noisy", "noise-free"])
@pytest.mark.parametrize("B,N,iters", [(64, 1024, 150), (5, 300, 64), (3, 2500, 200), (40, 64, 150), (7, 129, 150), (6, 700, 300)])
def test_split_form_equals_single_launch(B, N, iters, noise):
    """lc_pnp_ransac_init3_f32 (three launches, point chunks spread over the chip) against the one-workgroup-per-pose launch: same
    hypothesis stream, same per-point arithmetic, the same integers in the inlier counts AND the same float in the inlier error
    (both forms add it as even / odd sums per 64-point chunk, chunks in order), so the (count, error, id) arg-max picks the same
    hypothesis and an object gets the same initial pose whichever form its batch size selects -- also on noise-free data, where
    every hypothesis has every inlier and the winner is decided by the error sums alone."""
    from lc_amd import synth
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N, outlier_frac=0.3 if noise else 0.0, noise_px=noise).items()}
    g = torch.Generator().manual_seed(B)
    counts = torch.randint(max(4, N // 2), N + 1, (B,), generator=g).to(torch.int32)
    counts[0] = 3  # too few -> invalid in both forms
    outs = [gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], counts, reprojectionError=2.0, iterations=iters, seed=11,
                                    refine=False, return_hypothesis=True, split=s) for s in (True, False)]
    (st_a, in_a, bad_a, hyp_a, n_a), (st_b, in_b, bad_b, hyp_b, n_b) = outs
    assert torch.equal(bad_a, bad_b) and bool(bad_a[0])
    assert torch.equal(hyp_a, hyp_b), (hyp_a != hyp_b).nonzero().flatten().tolist()
    assert torch.equal(in_a, in_b) and torch.equal(n_a, n_b) and torch.equal(st_a, st_b)
