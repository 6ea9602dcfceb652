"""`lib.pnp.cer_solver.solve` call surface (`lib/pnp/cer_solver.py:6-53`): torch-facing weighted PnP.

Same arguments, defaults and return value as the reference -- `(invalid_dict, states)` with the keys 'solver_invalids'
and 'invalids', failed solves falling back to `start` -- but device-first: whatever arrives (batched tensors, or the
ragged per-job lists `test.py` builds) becomes one padded device batch + a device `n_points` vector and is solved by ONE
kernel launch; the reference's device -> host -> Ceres -> device trip (pnp_ceres.py:50-67, cer_solver.py:46-47) and its
per-job Python padding loop (cer_solver.py:67-87) do not exist here.
"""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn.utils.rnn import pad_sequence

from . import pnp_ceres


def _is_ragged(x) -> bool:
    return isinstance(x, (list, tuple))


def _pad(rows, device) -> Tensor:
    """Per-job tensors of different lengths -> one zero-padded (B, Nmax, ...) batch; per-job scalars/vectors -> (B, ...)."""
    if isinstance(rows, Tensor):
        return rows.to(device)
    first = rows[0]
    if not isinstance(first, Tensor):
        return torch.as_tensor(rows, device=device)
    rows = [r.to(device) for r in rows]
    if all(r.shape == first.shape for r in rows):
        return torch.stack(rows)
    return pad_sequence(rows, batch_first=True)


def _batch_tensors(*groups, device=None):
    """Batched view of every argument of a ragged call (the role of cer_solver.py:67-87); None stays None."""
    return [None if g is None else _pad(g, device) for g in groups]


def _information_factor(pts2d: Tensor, icovs: Tensor) -> Tensor:
    """What the kernel weighs residuals with: sqrt of a diagonal inverse covariance (B,N,2), or the lower Cholesky factor
    of a full one (B,N,2,2) -- cer_solver.py:33-38."""
    if icovs.dim() == pts2d.dim():
        return icovs.sqrt()
    return torch.linalg.cholesky_ex(icovs).L


@torch.no_grad()
def _weighted_solve(cam_mat, pts3d, pts2d, icovs, start, n_points, max_iter_count, num_workers, kwargs):
    states, _radius, flags = pnp_ceres.solve(cam_mat, pts3d, pts2d, _information_factor(pts2d, icovs), start, n_points,
                                             max_iter_count=max_iter_count, num_workers=num_workers, **kwargs)
    return states, flags


def solve(cam_mat, pts3d, pts2d, icovs, start, n_points=None, *, optimal_start=False, max_iter_count=50, num_workers=1,
          filter_input_nan=False, **kwargs):
    dev = pts3d[0].device
    if _is_ragged(pts3d):
        if n_points is None:
            n_points = [len(p) for p in pts3d]
        cam_mat, pts3d, pts2d, icovs, start, n_points = _batch_tensors(cam_mat, pts3d, pts2d, icovs, start, n_points, device=dev)
    elif _is_ragged(start):
        start = _pad(start, dev)
    # Device batches with a diagonal inverse covariance take the fused route: nan_to_num, the square root of the weights and the
    # fall-back to `start` happen inside the solver launch (lc_pnp_lm3_f32) instead of ~9 element-wise launches around it.
    fused = (not optimal_start and isinstance(pts3d, Tensor) and pts3d.is_cuda and isinstance(start, Tensor) and start.dim() == 2
             and isinstance(icovs, Tensor) and icovs.dim() == pts2d.dim() and not kwargs.get("print_summary", 0))
    if fused:
        states, _radius, flags = pnp_ceres.solve_device(cam_mat, pts3d, pts2d, icovs, start.detach(), n_points, max_iter_count=max_iter_count,
                                                        function_tolerance=kwargs.get("function_tolerance", 1e-6), weights_are_icov=True,
                                                        nan_to_num=bool(filter_input_nan))
        invalids = flags != 0
        return {"solver_invalids": invalids, "invalids": invalids}, states  # an invalid job already holds its (filtered) start
    if filter_input_nan:
        cam_mat, pts3d, pts2d, icovs, start = (torch.nan_to_num(t) for t in (cam_mat, pts3d, pts2d, icovs, start))
    start = start.detach()

    invalid_dict = {}
    if optimal_start:
        states = start
        invalids = torch.zeros(start.shape[:-1], dtype=torch.bool, device=start.device)
    else:
        states, flags = _weighted_solve(cam_mat, pts3d, pts2d, icovs, start, n_points, max_iter_count, num_workers, kwargs)
        states = states.to(dev)
        invalids = flags.to(device=dev, dtype=torch.bool)
        invalid_dict["solver_invalids"] = invalids
        states = torch.where(invalids[..., None], start.to(states.device), states)
    invalid_dict["invalids"] = invalids
    return invalid_dict, states
