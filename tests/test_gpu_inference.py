"""End-to-end test-time path (test.py:47-136) on the GPU: synthetic network outputs -> poses close to the ground truth."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def pose_err(a, b):
    qa = a[:, :4] / np.linalg.norm(a[:, :4], axis=1, keepdims=True)
    qb = b[:, :4] / np.linalg.norm(b[:, :4], axis=1, keepdims=True)
    sgn = np.sign((qa * qb).sum(1, keepdims=True))
    return np.abs(qa - sgn * qb).max(1), np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1) / np.linalg.norm(b[:, 4:], axis=1)


def test_sparse_inference_pipeline():
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp

    dev = torch.device("cuda:0")
    b = synth.make_batch(32, 16, seed=9, noise_px=0.5, outlier_frac=0.0)
    out = dict(pts2d=b["pts2d"].to(dev), pts2d_std=(1 / b["inv_std"]).to(dev))
    gt = dict(out_K=b["K"].to(dev), pts3d=b["pts3d"].to(dev))
    res = solve_pnp(AttrDict(solvers=["ransac", "weighted"]), out, gt)
    assert list(res) == ["weighted", "ransac"]
    dq, dt = pose_err(res["weighted"].cpu().numpy(), b["pose"].numpy())
    assert np.median(dq) < 2e-2 and dq.max() < 0.2 and np.median(dt) < 5e-2


@pytest.mark.parametrize("select", ["mask", "quantile", "quantile_in_mask"])
def test_dense_inference_pipeline(select):
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp
    from tests.golden.gen_golden_lossfn import dense_inputs

    dev = torch.device("cuda:0")
    gt, out = dense_inputs(B=4, H=32, W=32, seed=3)
    # make the weight logits favour the visible region so that quantile selection has signal
    out["xyz_weight_logits"] = out["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
    out["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
    gt = {k: v.to(dev) for k, v in gt.items()}
    out = {k: v.to(dev) for k, v in out.items()}
    np.random.seed(0)
    cfg = AttrDict(dense_point_select=select, quantile=0.5, dense_sample=2, solvers=["ransac", "weighted", "weighted_filtered"])
    res = solve_pnp(cfg, out, gt)
    assert list(res) == ["ransac", "weighted-filtered", "weighted"]
    for key in ("weighted", "weighted-filtered"):
        dq, dt = pose_err(res[key].cpu().numpy(), gt["pose_best"].cpu().numpy())
        assert dq.max() < 0.05 and dt.max() < 0.05, (key, dq, dt)


def test_graphed_dense_pipeline_replays_on_new_inputs():
    """The sync-free pipeline is hipGraph-capturable: a replay on NEW inputs equals the eager call on those inputs."""
    from lc_amd.config import AttrDict
    from lc_amd.inference import GraphedSolvePnP, solve_pnp
    from tests.golden.gen_golden_lossfn import dense_inputs

    dev = torch.device("cuda:0")

    def inputs(seed):
        gt, out = dense_inputs(B=4, H=32, W=32, seed=seed)
        out["xyz_weight_logits"] = out["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
        out["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
        return {k: v.to(dev) for k, v in gt.items()}, {k: v.to(dev) for k, v in out.items()}

    cfg = AttrDict(dense_point_select="quantile_in_mask", quantile=0.5, dense_sample=2, solvers=["ransac", "weighted", "weighted_filtered"])
    gt0, out0 = inputs(3)
    solver = GraphedSolvePnP(cfg, out0, gt0)
    for seed in (3, 4, 5):
        gt, out = inputs(seed)
        got, ref = solver(out, gt), solve_pnp(cfg, out, gt)
        assert list(got) == list(ref)
        assert all(torch.equal(got[k], ref[k]) for k in ref)
    with pytest.raises(ValueError):
        bad = dict(out, xyz_noc=out["xyz_noc"][:2])
        solver(bad, gt)


def test_sparse_pipeline_one_launch_solves_equal_the_separate_calls():
    """The sparse head's chain (test.py:47-64): the RANSAC's inlier refinement and the weighted solve as ONE launch, with `1 / std**2` formed at the solve's
    loads (LC_PNP_WEIGHTS_ARE_STD), against the separate calls on `std.pow(-2)`: the same poses bit for bit -- incl. a keypoint with a NaN deviation."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.inference import _weighted, solve_pnp
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    b = synth.make_batch(64, 16, seed=9, noise_px=0.5, outlier_frac=0.05)
    std = (1 / b["inv_std"]).to(dev)
    std[3, 5, 0] = float("nan")
    std[4, 2, 1] = float("inf")
    out = dict(pts2d=b["pts2d"].to(dev), pts2d_std=std)
    gt = dict(out_K=b["K"].to(dev), pts3d=b["pts3d"].to(dev))
    res = solve_pnp(AttrDict(solvers=["ransac", "weighted"]), out, gt)
    start, _inl, _bad = gpu_solver.solve_device(gt["out_K"], gt["pts3d"], out["pts2d"], reprojectionError=2)
    want = _weighted(gt["out_K"], gt["pts3d"], out["pts2d"], std.pow(-2), start)
    assert torch.equal(res["ransac"], start)
    assert torch.equal(res["weighted"], want)
