"""CPU oracle of the dense front end (TEST INFRASTRUCTURE ONLY): the reference's own op sequence, `losses.py:355-356`
(joint softmax x scale) followed by `losses.py:142-161 dense_pnp_matching_from_xyz`, in plain torch.  Pinned through the
`lossfn_dense_*` golden trajectories (tests/test_host_logic.py), which run this exact sequence inside Loss_fn."""
import torch


def gen_uv(H, W, dtype):
    xs = torch.arange(0, W - 0.5, dtype=dtype)
    ys = torch.arange(0, H - 0.5, dtype=dtype)
    x, y = torch.meshgrid((xs, ys), indexing="xy")
    return torch.stack((x, y), dim=-1)


def dense_front_end(xyz_noc, weight_logits, weights_scale, noc_scale, sample, top_left):
    top, left = top_left
    raw = weight_logits.reshape(weight_logits.shape[:-3] + (1, -1)).softmax(dim=-1)
    weights = raw.reshape_as(weight_logits) * weights_scale.reshape(-1, 1, 1, 1)
    H, W = weight_logits.shape[-2:]
    uv = gen_uv(H, W, weight_logits.dtype)
    pts2d = uv[top::sample, left::sample, :].flatten(0, 1)
    inv_std = weights[..., top::sample, left::sample].flatten(start_dim=-2).mT
    pts3d = None
    if xyz_noc is not None:  # binary-code heads decode their points elsewhere (losses.py:163-184)
        pts3d = xyz_noc[..., top::sample, left::sample].flatten(start_dim=-2).mT
        if noc_scale is not None:
            pts3d = pts3d * noc_scale.unsqueeze(-2)
    return pts2d.expand_as(inv_std), inv_std, pts3d
