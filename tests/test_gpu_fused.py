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
    go = torch.rand(B, device=dev) + 0.5
    unit = PoseUnit(B, N, dev)(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"], grad_out=go)
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
    st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    torch.cuda.synchronize()
    for a, c in ((unit.loss, loss), (unit.d_pts2d, du), (unit.d_inv_std, ds), (unit.d_pts3d, dx), (unit.states, st),
                 (unit.trust_radius, tr), (unit.invalid, ret)):
        assert torch.equal(a, c)


def test_pose_unit_rejects_large_n():
    from lc_amd.fused import PoseUnit

    with pytest.raises(ValueError):
        PoseUnit(2, 65, torch.device("cuda:0"))


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
