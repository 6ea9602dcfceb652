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
