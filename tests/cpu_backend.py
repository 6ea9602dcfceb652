"""Test-only backend: lets the HOST logic of lc_amd (autograd wiring, Loss_fn blending, cer_solver batching, sharding)
run on CPU by swapping the three launch functions for the oracle.  The product code never does this."""
import numpy as np
import pytest
import torch

from oracle import lc_loss_oracle, pnp_oracle


def _launch_loss(K, pose, pts3d, pts2d, inv_std, valid, bbox, grad_out, max_err_len, rel_thresh, w_e_thresh, want_grads, want_pts3d,
                 want_aux=False, cov_2d=False):
    with torch.enable_grad():  # we are called from inside autograd.Function.forward (grad mode off)
        loss, du, ds, dx = lc_loss_oracle.loss_and_grads(K, pose, pts3d, pts2d, inv_std, valid, bbox, grad_out=grad_out,
                                                         want_pts3d=want_pts3d, max_err_len=max_err_len, rel_thresh=rel_thresh,
                                                         w_e_thresh=w_e_thresh, cov_2d=cov_2d)
    if not want_grads:
        du = ds = dx = None
    return loss, du, ds, dx, None


def _launch_kpt(K, pose, pts3d, pts2d, std, want_grads):
    from oracle import kpt_oracle

    with torch.enable_grad():
        nll, du, ds = kpt_oracle.nll_and_grads(K, pose, pts3d, pts2d, std)
    return nll, (du if want_grads else None), (ds if want_grads else None)


def _launch_sqnorm(grads, workspace, state):
    from oracle import grad_oracle

    return grad_oracle.sum_of_squares(grads), state.detach().clone()


def _launch_apply(grads, sq, state_before, state, initial_max_norm, scale, momentum):
    from oracle import grad_oracle

    outs, new_state, norm = grad_oracle.apply(grads, sq, state_before.to(sq.dtype), initial_max_norm, scale, momentum)
    if state.dtype != new_state.dtype:  # the tests also run the host logic in fp64
        state.data = state.data.to(new_state.dtype)
    state.copy_(new_state)
    return outs, norm


def _launch_scale(scale, srcs):
    return [None if s is None else s * scale.view(-1, *([1] * (s.dim() - 1))) for s in srcs]


def _solve_device(cam_mat, pts3d, pts2d, sqrtL, start, n_points=None, *, max_iter_count=50, function_tolerance=1e-6, return_iters=False):
    L = sqrtL if sqrtL.dim() == 4 else torch.diag_embed(sqrtL)
    counts = None if n_points is None else np.asarray(torch.as_tensor(n_points).cpu(), np.int32)
    st, tr, ret = pnp_oracle.solve_batched(start.float().numpy(), cam_mat.float().numpy(), pts2d.float().numpy(), pts3d.float().numpy(),
                                           L.float().numpy(), counts=counts, max_iter=max_iter_count, ftol=function_tolerance)
    return torch.from_numpy(st), torch.from_numpy(tr), torch.from_numpy(ret)


def _dense_front_end(xyz_noc, wl, ws, noc_scale=None, sample=2, top_left=None):
    from oracle import dense_oracle

    top_left = tuple(np.random.randint(0, sample, size=2)) if top_left is None else top_left
    return dense_oracle.dense_front_end(xyz_noc, wl, ws, noc_scale, sample, top_left)


def _decode_with_gt_strided(logits, gt_raw_bits, bit_cnt, gt_msk, sample=1, top_left=(0, 0)):
    from oracle import floatbits_oracle

    t, l = top_left
    bits = list(bit_cnt) if isinstance(bit_cnt, (list, tuple)) else [bit_cnt] * 3
    from lc_amd import floatbits
    noc = floatbits_oracle.nn_logits2noc_with_gt(logits[..., t::sample, l::sample], gt_raw_bits[..., t::sample, l::sample], bits,
                                                 gt_msk[..., t::sample, l::sample], black=floatbits._black_background)
    return noc.flatten(1, 2)


def apply(setter):
    """Swap every launch function of lc_amd for its oracle twin through `setter(obj, name, value)` (pytest's
    monkeypatch.setattr in the fixture below; plain setattr in spawned children and one-off subprocesses)."""
    from lc_amd import _lib, cov_mixed, grad, kpt, losses
    from lc_amd.pnp import pnp_ceres

    setter(losses, "dense_front_end", _dense_front_end)
    setter(losses.floatbits, "decode_with_gt_strided", _decode_with_gt_strided)
    setter(_lib, "require_hip_f32", lambda name, t: t.contiguous())
    setter(_lib, "require_hip_map", lambda name, t: t.contiguous())
    setter(cov_mixed, "_launch_loss", _launch_loss)
    setter(cov_mixed, "_launch_scale", _launch_scale)
    setter(kpt, "_launch_kpt", _launch_kpt)
    setter(grad, "_launch_sqnorm", _launch_sqnorm)
    setter(grad, "_launch_apply", _launch_apply)
    setter(grad.NormClipper, "_ws", lambda self, dev: None)
    setter(pnp_ceres, "solve_device", _solve_device)
    orig_solve = pnp_ceres.solve

    def solve(cam_mat, pts3d, pts2d, sqrtL, start, n_points=None, *, max_iter_count=50, num_workers=1, **kw):
        if isinstance(pts3d, torch.Tensor) and isinstance(start, torch.Tensor) and start.dim() == 2:
            return _solve_device(cam_mat, pts3d, pts2d, sqrtL, start, n_points, max_iter_count=max_iter_count,
                                 function_tolerance=kw.get("function_tolerance", 1e-6))
        return orig_solve(cam_mat, pts3d, pts2d, sqrtL, start, n_points, max_iter_count=max_iter_count, num_workers=num_workers, **kw)

    setter(pnp_ceres, "solve", solve)


@pytest.fixture
def oracle_backend(monkeypatch):
    apply(monkeypatch.setattr)
    yield
