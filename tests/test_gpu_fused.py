"""The one-launch pose unit must reproduce the two stand-alone kernels bit for bit (same device code, shared grid)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N", [(256, 64), (5, 17), (1, 3), (1000, 40)])
def test_pose_unit_equals_separate_kernels(B, N):
    from lc_amd import synth
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    from lc_amd.fused import PoseUnit
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N).items()}
    go = (torch.rand(B, generator=torch.Generator().manual_seed(B + N)) + 0.5).to(dev)  # seeded: a test's inputs are part of the test
    unit = PoseUnit(B, N, dev)(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"], grad_out=go)
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
    st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    torch.cuda.synchronize()
    for a, c in ((unit.loss, loss), (unit.d_pts2d, du), (unit.d_inv_std, ds), (unit.d_pts3d, dx), (unit.states, st),
                 (unit.trust_radius, tr), (unit.invalid, ret)):
        assert torch.equal(a, c)


def test_pose_unit_rejects_shapes_without_a_fused_form():
    from lc_amd.fused import PoseUnit

    for B, N in ((2, 65), (2, 256), (300, 1024), (2, 4096)):  # between the forms; more loss workgroups than the tiled form takes; too wide
        with pytest.raises(ValueError):
            PoseUnit(B, N, torch.device("cuda:0"))


@pytest.mark.parametrize("B,N", [(32, 1024), (32, 1849), (5, 700), (64, 1024), (3, 2048)])
def test_dense_pose_unit_equals_separate_kernels(B, N):
    """lc_pose_unit2_f32 for the dense shapes (tiled loss workgroups + four-wave solve workgroups in one grid) against the two
    stand-alone launches: loss, gradients, poses, radii, flags, iteration counts bit for bit -- twice in a row (the workspace is left
    as found)."""
    from lc_amd import synth
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    from lc_amd.fused import PoseUnit
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N).items()}
    go = (torch.rand(B, generator=torch.Generator().manual_seed(B + N)) + 0.5).to(dev)  # seeded: a test's inputs are part of the test
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
    st, tr, ret, it = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], return_iters=True)
    unit = PoseUnit(B, N, dev)
    for _ in range(2):
        for t in (unit.loss, unit.d_pts2d, unit.d_inv_std, unit.d_pts3d, unit.states):
            t.fill_(float("nan"))
        unit(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"], grad_out=go)
        torch.cuda.synchronize()
        for a, c in ((unit.loss, loss), (unit.d_pts2d, du), (unit.d_inv_std, ds), (unit.d_pts3d, dx), (unit.states, st),
                     (unit.trust_radius, tr), (unit.invalid, ret), (unit.iters, it)):
            assert torch.equal(a, c)
        assert int(unit.ws[:4 * (4 + 2 * B)].count_nonzero()) == 0  # the hand-off's counters are back at zero (the tile rows behind them are scratch)


def test_pose_unit_options_equal_separate_kernels():
    """valid mask, no d_pts3d, non-default LM settings and loss knobs go through the shared grid unchanged."""
    from lc_amd import synth
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    from lc_amd.fused import PoseUnit
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    B, N = 37, 48
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=11, outlier_frac=0.2).items()}
    g = torch.Generator().manual_seed(4)
    valid = (torch.rand(B, N, generator=g) > 0.3).float().to(dev)
    unit = PoseUnit(B, N, dev, want_pts3d=False)(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"],
                                                 valid=valid, max_iter_count=4, function_tolerance=1e-9, max_err_len=10, rel_thresh=2,
                                                 w_e_thresh=3)
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], valid, b["bbox_3d"],
                                               want_pts3d=False, max_err_len=10, rel_thresh=2, w_e_thresh=3)
    st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], max_iter_count=4,
                                         function_tolerance=1e-9)
    torch.cuda.synchronize()
    assert dx is None and unit.d_pts3d is None and int(ret.sum()) > 0  # four iterations at ftol 1e-9: some jobs end NO_CONVERGENCE
    for a, c in ((unit.loss, loss), (unit.d_pts2d, du), (unit.d_inv_std, ds), (unit.states, st), (unit.trust_radius, tr), (unit.invalid, ret)):
        assert torch.equal(a, c)


def test_pose_unit_vs_oracle_and_reference_golden():
    """The headline launch itself (not via the stand-alone kernels) against (a) the reference's own outputs for the metric
    configuration -- golden lc_loss_metric_B256_N64 generated by the unmodified lib.cov_mixed.Loss_cov_mixed -- and (b) the
    oracles: LC loss + gradients in fp64, weighted-PnP poses / flags / trust radii."""
    import os

    import numpy as np

    from lc_amd.fused import PoseUnit
    from oracle import pnp_oracle
    from tests.pnp_cases import pose_err
    from tests.util import GOLDEN, load_loss_case, rel_err

    dev = torch.device("cuda:0")
    z, ins, kwargs, want_x = load_loss_case(os.path.join(GOLDEN, "lc_loss_metric_B256_N64.npz"), torch.float32)
    B, N = ins["pts3d"].shape[:2]
    d = {k: v.to(dev) for k, v in ins.items()}  # K, pose, pts3d, pts2d, inv_std, bbox_3d, start (the metric's perturbed start poses), grad_out
    unit = PoseUnit(B, N, dev)(d["K"], d["pose"], d["pts3d"], d["pts2d"], d["inv_std"], d["bbox_3d"], d["start"], grad_out=d["grad_out"])
    torch.cuda.synchronize()
    # (a) reference outputs (fp64 run of the unmodified reference on the same inputs)
    ref = torch.from_numpy(z["f64_loss"]).double()
    assert ((unit.loss.cpu().double() - ref).abs() / ref.abs().clamp_min(1)).max().item() <= 3e-5
    assert rel_err(unit.d_pts2d.cpu(), z["f64_g_pts2d"]) <= 3e-4 and rel_err(unit.d_inv_std.cpu(), z["f64_g_inv_std"]) <= 3e-4
    # (b) the PnP half against the oracle on the same correspondences
    so, tro, reto = pnp_oracle.solve_batched(ins["start"].numpy(), ins["K"].numpy(), ins["pts2d"].numpy(), ins["pts3d"].numpy(),
                                             torch.diag_embed(ins["inv_std"]).numpy(), num_threads=4)
    np.testing.assert_array_equal(unit.invalid.cpu().numpy(), reto)
    dq, dt = pose_err(unit.states.cpu().numpy(), so)
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4
    np.testing.assert_allclose(unit.trust_radius.cpu().numpy(), tro, rtol=1e-6)
