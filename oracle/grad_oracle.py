"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's adaptive gradient clipping:
  clip_norm      lib/utils/grad.py:33-83   total 2-norm over all tensors, coefficient max_norm/(norm+1e-6) clamped at 1
  NormClipper    lib/utils/grad.py:5-30    first call(s): limit = initial_max_norm, max_norm <- norm*scale;
                                           afterwards: limit = max_norm, max_norm <- max_norm*(1-m) + m*scale*min(norm, max_norm*scale)
Pinned through the `Loss_fn` dense goldens (tests/golden/lossfn_dense_*.npz hold the reference's max_norm trajectory and
the clipped gradients) and tests/test_oracle_grad.py (the reference class imported in this container).
"""
import torch


def sum_of_squares(grads):
    return torch.stack([g.detach().pow(2).sum() for g in grads]).sum()


def apply(grads, sq, state, initial_max_norm, scale, momentum):
    """-> (clipped list, new state, norm) from the (possibly all-reduced) sum of squares."""
    norm = sq.sqrt()
    fresh = bool(state <= 0)
    limit = initial_max_norm if fresh else state
    coef = torch.clamp(limit / (norm + 1e-6), max=1.0)
    new_state = norm * scale if fresh else state * (1 - momentum) + momentum * scale * norm.clamp_max(state * scale)
    return [g * coef for g in grads], new_state, norm
