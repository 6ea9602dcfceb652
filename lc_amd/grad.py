"""`lib.utils.grad.NormClipper` (`lib/utils/grad.py:5-30`): EMA-adaptive global-norm clipping used as a backward hook
on the weight logits (`losses.py:245-247,343-352`).  Buffer name `max_norm` is kept so checkpoints load strictly.

Under batch sharding the norm must be taken over the WHOLE batch gradient (grad.py:66-68): pass a process group and
the squared norm is all-reduced (one float over RCCL) before the clip coefficient is formed.
"""
from __future__ import annotations

import torch
from torch import Tensor


def clip_norm(grads, max_norm, norm_type: float = 2.0, group=None):
    if isinstance(grads, Tensor):
        grads = [grads]
    norm_type = float(norm_type)
    if len(grads) == 0:
        return torch.tensor(0.0), []
    if norm_type != 2.0:
        raise NotImplementedError("lc_amd NormClipper: only the 2-norm the reference uses is implemented")
    sq = torch.stack([p.detach().pow(2).sum() for p in grads]).sum()
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=group)
    total_norm = sq.sqrt()
    clip_coef = max_norm / (total_norm + 1e-6)
    clip_coef_clamped = torch.clamp(clip_coef, max=1.0)
    clipped = [p.mul(clip_coef_clamped.to(p.device)) for p in grads]
    return total_norm, clipped


class NormClipper(torch.nn.Module):
    def __init__(self, initial_max_norm=100, rel_thresh=0.7, momentum=0.1, group=None) -> None:
        super().__init__()
        self.initial_max_norm = initial_max_norm
        self.register_buffer("max_norm", torch.tensor(-1, dtype=torch.float))
        self.momentum = momentum
        self.scale = 1 + rel_thresh
        self.last_norm = 0
        self.start = True
        self.group = group

    def forward(self, grads, norm_type=2):
        return self.clip(grads, norm_type)

    def clip(self, grads, norm_type=2):
        if self.start and self.max_norm <= 0:
            new_norm, clipped = clip_norm(grads, self.initial_max_norm, norm_type=norm_type, group=self.group)
            self.max_norm = new_norm * self.scale
        else:
            self.start = False
            new_norm, clipped = clip_norm(grads, self.max_norm, norm_type=norm_type, group=self.group)
            self.max_norm = self.max_norm * (1 - self.momentum) \
                + self.momentum * self.scale * new_norm.clamp_max(self.max_norm * self.scale)
        self.last_norm = new_norm
        return clipped[0] if isinstance(grads, Tensor) else clipped
