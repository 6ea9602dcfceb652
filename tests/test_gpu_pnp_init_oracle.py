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
    # the winner's inlier mask is exact at every point the oracle does not see within 1e-3 of the threshold (dense rows: a few of thousands)
    open_pts = z["mask_unsure"].astype(bool)
    assert (inl == z["inlier_mask"].astype(bool))[decided][~open_pts[decided]].all()
    assert (np.abs(n_in - z["n_inliers"])[decided] <= open_pts.sum(1)[decided]).all()
    dq, dt = pose_err(st[decided], z["states"][decided]) if decided.any() else (np.zeros(1), np.zeros(1))
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (dq.max(), dt.max())  # two different P3P algorithms, same minimal sample
    # poses the oracle leaves open (ties / points on the threshold): the kernel's winner is still one of the oracle's best
    cnt = z["per_hyp_count"].astype(np.int32)
    best = cnt.max(1)
    rows = np.nonzero(valid)[0]
    assert (cnt[rows, hyp[rows]] >= best[rows] - 2).all(), (cnt[rows, hyp[rows]], best[rows])
    if "tail" in name:  # the consensus lives behind the first 2500 points of every row: found only by sampling and scoring ALL points
        assert valid.all() and (n_in >= 0.7 * (~z["in_outlier"]).sum(1)).all() and (inl[:, :2500].sum(1) <= 50).all()  # chance hits only
        dq, dt = pose_err(st, z["in_pose_gt"])
        assert dq.max() < 0.1 and dt.max() < 0.1, (dq.max(), dt.max())  # an unrefined minimal-sample pose at 0.3 px noise: inside the LM basin
    if "clean" in name:  # noise-free: every point is an inlier of the winner, and the pose is the ground truth
        assert inl[valid].all() and (n_in[valid] == z["in_counts"][valid]).all()
        dq, dt = pose_err(st[valid], z["in_pose_gt"][valid])
        assert dq.max() < 2e-4 and dt.max() < 2e-4
    else:
        assert decided.sum() >= 0.6 * valid.sum()


@pytest.mark.parametrize("noise", [0.7, 0.0], ids=["noisy", "noise-free"])
@pytest.mark.parametrize("B,N,iters", [(64, 1024, 150), (5, 300, 64), (3, 2500, 200), (40, 64, 150), (7, 129, 150), (6, 700, 300),
                                       (4, 9000, 150), (2, 16384, 150), (3, 4097, 300),  # several LDS tiles / more than 32 chunks per pose
                                       (130, 4100, 64), (100, 5000, 200)])  # wide rows: more / fewer than 128 poses (the two scoring kernels of wide rows)
def test_split_form_equals_single_launch(B, N, iters, noise):
    """lc_pnp_ransac_init5_f32 with a workspace (three launches, point chunks spread over the chip) against the one-workgroup-per-pose launch: same
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
                                    refine=False, return_hypothesis=True, split=s, ticketed=t) for s, t in ((True, False), (False, False), (True, True))]
    (st_a, in_a, bad_a, hyp_a, n_a), (st_b, in_b, bad_b, hyp_b, n_b), ticketed = outs
    assert torch.equal(bad_a, bad_b) and bool(bad_a[0])
    assert torch.equal(hyp_a, hyp_b), (hyp_a != hyp_b).nonzero().flatten().tolist()
    assert torch.equal(in_a, in_b) and torch.equal(n_a, n_b) and torch.equal(st_a, st_b)
    # ticketed form: the selection inside the scoring launch (the last workgroup of a pose to finish selects) -- same outputs
    for x, y in zip(ticketed, outs[0]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("split,ticketed", [(True, False), (True, True), (False, False)], ids=["split", "split_ticketed", "single_launch"])
@pytest.mark.parametrize("B,N,iters,min_count", [(64, 1024, 150, 4), (5, 300, 64, 4), (3, 2500, 200, 6), (40, 64, 150, 4), (7, 129, 150, 4),
                                                 (3, 6000, 150, 4)])
def test_fused_inlier_reselection_equals_dense_select(B, N, iters, min_count, split, ticketed):
    """`select=` of lc_pnp_ransac_init5_f32 (the inliers compacted by the workgroup that writes the inlier mask) against
    lc_dense_select_f32 in 'mask' mode run on that mask in a launch of its own: rows, counts and source indices bit for bit -- incl. a
    pose with too few points, poses RANSAC gives up on (nothing kept: padded with min_count pseudo-random entries) and an input that
    is itself a compacted selection (index != identity)."""
    from lc_amd import synth
    from lc_amd.dense import dense_select
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=3 * B + N, outlier_frac=0.4, noise_px=0.7).items()}
    g = torch.Generator().manual_seed(N)
    counts = torch.randint(max(4, N // 2), N + 1, (B,), generator=g).to(torch.int32).to(dev)
    counts[0] = 3
    b["pts3d"][1] = 0  # a degenerate object: no hypothesis, RANSAC gives up, nothing is kept
    w = torch.rand(B, N, 2, generator=g).to(dev) + 0.1
    index = torch.stack([torch.randperm(4 * N, generator=g)[:N].sort().values for _ in range(B)]).to(torch.int32).to(dev)
    sel = dict(weights=w, index=index, min_count=min_count, seed=5)
    st, inl, bad = gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], counts, reprojectionError=2.0, iterations=iters, seed=11,
                                           refine=False, split=split, ticketed=ticketed, select=sel)
    assert bool(bad[0]) and bool(bad[1]) and not bool(bad[2:].all())
    want = dense_select(b["pts2d"], w, b["pts3d"], "mask", mask=inl, counts=counts, index=index, square_weights=False, min_count=min_count, seed=5)
    got = sel["result"]
    cnt = want[3]
    assert torch.equal(got[3], cnt) and int(cnt[0]) == 0 and int(cnt[1]) == min_count and int(cnt.max()) > min_count
    live = (torch.arange(N, device=dev)[None, :] < cnt[:, None])  # entries behind a row's count are undefined in both
    for name, x, y in zip(("pts2d", "weights", "pts3d", "", "index"), got, want):
        if name:
            m = live if x.dim() == 2 else live[..., None].expand_as(x)
            assert torch.equal(x[m], y[m]), name


def test_ticketed_form_under_concurrency_and_replay():
    """The arrival counters of the ticketed form: 200 back-to-back launches on two streams with workspaces of their own, and 50 replays
    of a captured launch, give the outputs of the three-launch form every time."""
    from lc_amd import synth
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    shapes = [(64, 1024), (33, 700)]
    data, want = [], []
    for B, N in shapes:
        b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B, outlier_frac=0.3, noise_px=0.5).items()}
        data.append(b)
        want.append(gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], None, reprojectionError=2.0, seed=3, refine=False,
                                            return_hypothesis=True, split=True, ticketed=False))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in shapes]
    got = [[] for _ in shapes]
    for it in range(100):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                b = data[k]
                got[k].append(gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], None, reprojectionError=2.0, seed=3, refine=False,
                                                      return_hypothesis=True, split=True, ticketed=True))
    torch.cuda.synchronize()
    for k in range(len(shapes)):
        for outs in got[k]:
            for x, y in zip(outs, want[k]):
                assert torch.equal(x, y)
    b = data[0]
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(2):
            gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], None, reprojectionError=2.0, seed=3, refine=False, split=True, ticketed=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            outs = gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], None, reprojectionError=2.0, seed=3, refine=False,
                                           return_hypothesis=True, split=True, ticketed=True)
    for _ in range(50):
        for t in outs:
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for x, y in zip(outs, want[0]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("split", [True, False], ids=["split", "single_launch"])
@pytest.mark.parametrize("B,N,cut", [(24, 64, 8), (10, 700, 3), (5, 5000, 2)])
def test_pose_index_offset_makes_sub_batches_equal_the_one_batch(B, N, cut, split):
    """lc_pnp_ransac_init5_f32: a slice [cut, B) of a batch solved with pose_index_offset = cut draws the hypothesis streams (and the padding of
    the inlier re-selection) of poses cut .. B-1 of the whole batch: states, masks, winners, re-selected rows bit for bit."""
    from lc_amd import synth
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B * N, outlier_frac=0.35, noise_px=0.6).items()}
    g = torch.Generator().manual_seed(N)
    counts = torch.randint(max(4, N // 2), N + 1, (B,), generator=g).to(torch.int32).to(dev)
    b["pts3d"][cut + 1] = 0  # a degenerate object in the slice: RANSAC gives up, the re-selection pads with pseudo-random entries keyed by the pose index
    w = (torch.rand(B, N, 2, generator=g) + 0.1).to(dev)

    def run(lo, hi, off):
        sel = dict(weights=w[lo:hi], min_count=4, seed=9)
        out = gpu_solver.solve_device(b["K"][lo:hi], b["pts3d"][lo:hi], b["pts2d"][lo:hi], counts[lo:hi], reprojectionError=2.0, seed=5, refine=False,
                                      return_hypothesis=True, split=split, select=sel, pose_index_offset=off)
        return out, sel["result"]

    (st, inl, bad, hyp, n_in), rows = run(0, B, 0)
    (st2, inl2, bad2, hyp2, n2), rows2 = run(cut, B, cut)
    assert bool(bad[cut + 1]) and torch.equal(bad[cut:], bad2)
    assert torch.equal(st[cut:], st2) and torch.equal(inl[cut:], inl2) and torch.equal(hyp[cut:], hyp2) and torch.equal(n_in[cut:], n2)
    cnt = rows[3][cut:]
    assert torch.equal(cnt, rows2[3])
    live = torch.arange(N, device=dev)[None, :] < cnt[:, None]
    for k in (0, 1, 2, 4):
        m = live if rows[k].dim() == 2 else live[..., None].expand_as(rows2[k])
        assert torch.equal(rows[k][cut:][m], rows2[k][m]), k
    # without the offset the slice is a batch of its own: other hypothesis streams (the winners differ somewhere)
    (_, _, _, hyp3, _), _ = run(cut, B, 0)
    assert not torch.equal(hyp3, hyp2)
