"""The weighted-PnP solve of few poses x thousands of correspondences, several workgroups per pose (`lc_pnp_lm3_f32` with a workspace,
lc_amd/csrc/lc_pnp.hip: lc_pnp_lm_split_kernel) against the one-workgroup-per-pose solve of the same inputs: the same LM schedule
(iteration counts, validity), poses equal up to the order of the fp64 sums; and against the CPU oracle (oracle/pnp_lm_oracle.c, the
restatement of ceres.cpp:16-147) like every other form of the solve."""
import numpy as np
import pytest
import torch

from lc_amd import _lib, synth
from lc_amd.pnp import pnp_ceres
from oracle import pnp_oracle

from .test_gpu_pnp import pose_err

pytestmark = pytest.mark.gpu
POSE_BYTES = 2 * 8 * 64 * 8 + 128  # lc_common.h kSplitPoseBytes: two rows of 8 x 32 partial sums (two ticketed words each) + the tail's line (epoch, dirty, rescues)


def _batch(B, N, seed, noise_px=0.7, outlier_frac=0.05):
    dev = torch.device("cuda:0")
    return {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=seed, outlier_frac=outlier_frac, noise_px=noise_px).items()}


def _ragged_counts(B, N, seed, low=3):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(low, N + 1, (B,), generator=g, dtype=torch.int32)
    c[0] = N
    if B > 2:
        c[1], c[2] = 2, 0  # fewer than three correspondences: invalid without a solve (ceres.cpp:84-91), on every part of the pose
    return c.cuda()


@pytest.mark.parametrize("B,N,parts", [(64, 4096, 4), (24, 3000, 8), (100, 2100, 2), (8, 16384, 8), (3, 2049, 8)])
def test_split_solve_equals_the_one_workgroup_solve(B, N, parts):
    lib = _lib.load()
    assert lib.lc_pnp_lm_workspace_bytes(B, N) == B * POSE_BYTES
    b = _batch(B, N, seed=B + N)
    counts = _ragged_counts(B, N, seed=N)
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts)
    one = pnp_ceres.solve_device(*args, return_iters=True, split=False)
    ws = pnp_ceres.split_workspace(torch.device("cuda:0"), (B, N))
    epoch_of = lambda: ws[:B * POSE_BYTES].view(torch.int32).view(B, POSE_BYTES // 4)[:, -32].clone()  # noqa: E731  (the per-stream workspace is shared with the tests before this one)
    torch.cuda.synchronize()
    epoch0 = epoch_of()
    for rep in range(3):  # the workspace is left ready: repeated launches on it give the same result
        many = pnp_ceres.solve_device(*args, return_iters=True, split=True)
        assert torch.equal(one[2], many[2]), "validity"
        assert torch.equal(one[3], many[3]), "LM iterations"
        ok = one[2] == 0
        assert ok.sum() >= B - 3
        np.testing.assert_allclose(many[0][ok].cpu().numpy(), one[0][ok].cpu().numpy(), rtol=0, atol=2e-6)
        assert torch.equal(many[0][~ok], one[0][~ok]), "invalid jobs return the start"
        np.testing.assert_allclose(many[1].cpu().numpy(), one[1].cpu().numpy(), rtol=1e-5)
    torch.cuda.synchronize()
    grown = epoch_of() - epoch0
    solved = counts >= 3
    assert bool((grown[solved] >= 3 * 2).all()) and bool((grown[~solved] == 0).all()), "the tickets of a pose's sums count on from launch to launch"


def test_shapes_outside_the_split_form_take_the_plain_kernel():
    lib = _lib.load()
    for B, N in ((64, 2048), (129, 4096), (256, 64), (0, 4096)):
        assert lib.lc_pnp_lm_workspace_bytes(B, N) == 0
    b = _batch(16, 1024, seed=5)
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    one, many = pnp_ceres.solve_device(*args, split=False), pnp_ceres.solve_device(*args, split=True)
    assert all(torch.equal(a, c) for a, c in zip(one, many))


def test_split_solve_with_load_time_options_and_shared_poses():
    """The forms the test-time chain uses: unit weights on a mask; inverse variances + nan_to_num with the start shared by two selections."""
    B, N = 32, 2500
    b = _batch(B, N, seed=11)
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(B, N, generator=g) < 0.7).cuda()
    counts = _ragged_counts(B, N, seed=2, low=1500)
    for kw in (dict(weight_mask=mask), dict(weight_mask=mask, max_iter_count=20)):
        one = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], None, b["start"], counts, return_iters=True, split=False, **kw)
        many = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], None, b["start"], counts, return_iters=True, split=True, **kw)
        assert torch.equal(one[2], many[2]) and torch.equal(one[3], many[3])
        np.testing.assert_allclose(many[0].cpu().numpy(), one[0].cpu().numpy(), rtol=0, atol=2e-6)
    icov = b["inv_std"] ** 2
    icov[0, 5, 0] = float("nan")
    two = lambda t: torch.cat([t, t.flip(1)])  # noqa: E731  (two selections of the same objects: the same points in another order)
    args = (b["K"], two(b["pts3d"]), two(b["pts2d"]), two(icov), b["start"], torch.cat([counts, torch.full_like(counts, N)]))
    kw = dict(weights_are_icov=True, nan_to_num=True, shared_poses=B, return_iters=True)
    one, many = pnp_ceres.solve_device(*args, split=False, **kw), pnp_ceres.solve_device(*args, split=True, **kw)
    assert torch.equal(one[2], many[2]) and torch.equal(one[3], many[3])
    np.testing.assert_allclose(many[0].cpu().numpy(), one[0].cpu().numpy(), rtol=0, atol=2e-6)


def test_split_solve_against_the_cpu_oracle():
    B, N = 6, 2600
    b = _batch(B, N, seed=21, outlier_frac=0.0)
    counts = torch.tensor([N, 2300, 2100, N, 2555, 2049], dtype=torch.int32).cuda()
    state, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts, split=True)
    cpu = {k: v.cpu().numpy() for k, v in b.items()}
    so, tro, reto = pnp_oracle.solve_batched(cpu["start"], cpu["K"], cpu["pts2d"], cpu["pts3d"], torch.diag_embed(b["inv_std"]).cpu().numpy(),
                                             counts=counts.cpu().numpy(), num_threads=4)
    np.testing.assert_array_equal(ret.cpu().numpy(), reto)
    dq, dt = pose_err(state.cpu().numpy(), so)
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (dq.max(), dt.max())
    assert np.allclose(tr.cpu().numpy(), tro, rtol=1e-3)


def test_chain_call_with_a_workspace_equals_the_two_split_solves():
    B, N = 64, 3000
    b = _batch(B, N, seed=31)
    g = torch.Generator().manual_seed(4)
    mask = (torch.rand(B, N, generator=g) < 0.8).cuda()
    counts = _ragged_counts(B, N, seed=9, low=2000)
    first = dict(cam_mat=b["K"], pts3d=b["pts3d"], pts2d=b["pts2d"], sqrtL=None, start=b["start"], n_points=counts, weight_mask=mask, max_iter_count=20)
    icov = b["inv_std"] ** 2
    second = dict(cam_mat=b["K"], pts3d=b["pts3d"], pts2d=b["pts2d"], sqrtL=icov, n_points=counts, weights_are_icov=True, nan_to_num=True)
    (s1, _, r1), (s2, _, r2) = pnp_ceres.solve_chain_device(first, dict(second, start="first"), split=True)
    e1 = pnp_ceres.solve_device(**first, split=True)
    e2 = pnp_ceres.solve_device(**dict(second, start=e1[0]), split=True)
    assert torch.equal(s1, e1[0]) and torch.equal(r1, e1[2]) and torch.equal(s2, e2[0]) and torch.equal(r2, e2[2])


@pytest.mark.timeout(180)
def test_split_launches_side_by_side_lose_no_pose():
    """Split solves launched on TWO streams at once (each with a workspace of its own).  A split launch is sized to one workgroup per compute
    unit, so two of them can hold units the other's missing workgroups need; a part's wait is bounded and the rescue launch behind every
    split launch re-solves what its parts gave up on (tests/test_gpu_contention.py has the forced case).  The contract: every call returns the
    one-launch-at-a-time result bit for bit -- no pose is reported invalid because of scheduling -- and the workspaces stay usable."""
    B, N = 64, 4096
    b = _batch(B, N, seed=41)
    counts = torch.full((B,), 3000, dtype=torch.int32, device="cuda:0")
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts)
    want = pnp_ceres.solve_device(*args, split=True)
    torch.cuda.synchronize()
    assert int(want[2].sum()) == 0
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(6):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[k].append(pnp_ceres.solve_device(*args, split=True))
    torch.cuda.synchronize()
    for per_stream in outs:
        for state, tr, ret in per_stream:
            assert torch.equal(ret, want[2]) and torch.equal(state, want[0]) and torch.equal(tr, want[1])
    for st in streams:  # the workspaces survive: an undisturbed solve on either stream is exact again
        with torch.cuda.stream(st):
            again = pnp_ceres.solve_device(*args, split=True)
        st.synchronize()
        assert torch.equal(again[0], want[0]) and int(again[2].sum()) == 0


def test_split_solve_soak_over_random_shapes():
    """Forty random batches (poses, row length, ragged counts incl. poses below three points, weight forms): the split solve keeps the LM schedule of the
    one-workgroup solve pose for pose -- same validity flags, same iteration counts -- and the poses within 2e-6."""
    g = torch.Generator().manual_seed(123)
    tried = 0
    for case in range(40):
        B = int(torch.randint(1, 129, (1,), generator=g))
        N = int(torch.randint(2049, 9000, (1,), generator=g))
        if _lib.load().lc_pnp_lm_workspace_bytes(B, N) == 0:
            continue
        tried += 1
        b = _batch(B, N, seed=1000 + case, noise_px=float(torch.rand(1, generator=g)) * 1.5, outlier_frac=0.1 * float(torch.rand(1, generator=g)))
        counts = torch.randint(0, N + 1, (B,), generator=g, dtype=torch.int32).cuda()
        kw = {}
        if case % 3 == 1:
            kw = dict(weight_mask=(torch.rand(B, N, generator=g) < 0.7).cuda(), max_iter_count=20)
        elif case % 3 == 2:
            kw = dict(weights_are_icov=True, nan_to_num=True)
        w = None if "weight_mask" in kw else (b["inv_std"] ** 2 if kw else b["inv_std"])
        one = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], w, b["start"], counts, return_iters=True, split=False, **kw)
        many = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], w, b["start"], counts, return_iters=True, split=True, **kw)
        assert torch.equal(one[2], many[2]) and torch.equal(one[3], many[3]), (case, B, N)
        np.testing.assert_allclose(many[0].cpu().numpy(), one[0].cpu().numpy(), rtol=0, atol=2e-6, err_msg=str((case, B, N)))
    assert tried >= 30


def test_owned_workspaces_are_sized_from_the_device_and_a_rejected_one_warns():
    """`GraphedSolvePnP` sized its split workspaces from constants for a 256-CU device and `splitws.get` silently fell back when an owned
    workspace did not fit.  Now `max_bytes` follows the device's compute units, devices are compared by index (torch.device('cuda') is the current
    device), and a rejected owned workspace warns before the fallback -- which still solves correctly."""
    import warnings

    from lc_amd import splitws

    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert splitws.max_bytes("pnp", dev) == (cus // 2) * POSE_BYTES and splitws.max_bytes("select") == (cus // 2) * splitws.SELECT_POSE_BYTES
    B, N = 16, 2500
    b = _batch(B, N, seed=3)
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    want = pnp_ceres.solve_device(*args, split=True)
    big = torch.zeros(splitws.max_bytes("pnp", dev), device=torch.device("cuda"), dtype=torch.uint8)  # no index: still this device
    with splitws.owned(pnp=big), warnings.catch_warnings():
        warnings.simplefilter("error")
        got = pnp_ceres.solve_device(*args, split=True)
    assert all(torch.equal(x, y) for x, y in zip(got, want)) and int(big.view(torch.int32).view(-1, POSE_BYTES // 4)[:B, -32].min()) > 0  # the owned one was used
    small = torch.zeros(B * POSE_BYTES - 128, device=dev, dtype=torch.uint8)
    with splitws.owned(pnp=small), pytest.warns(RuntimeWarning, match="does not serve"):
        got = pnp_ceres.solve_device(*args, split=True)
    assert all(torch.equal(x, y) for x, y in zip(got, want)) and int(small.sum()) == 0  # ... and the small one was not touched
