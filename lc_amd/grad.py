"""`lib.utils.grad.NormClipper` call surface (`lib/utils/grad.py:5-30`): EMA-adaptive global-norm clipping, hung as a backward
hook on the dense heads' weight logits / scale / points (`losses.py:245-247,343-352,378-381`).  The buffer name `max_norm`
is kept so checkpoints load strictly.

Two launches per hook call and no host synchronisation (`lc_amd/csrc/lc_clip.hip`): sum of squares -> device scalar, then
coefficient + scaling + the EMA update of `max_norm` from device scalars.  `max_norm` is updated IN PLACE (the first launch
snapshots it), so the addresses are fixed and a step containing the hook can be replayed as a hipGraph.  The reference's `start` flag only exists to
skip a device->host read of `max_norm <= 0`; with the test evaluated on the device it has no role (`max_norm` stays
positive once it has been set, so "not started" and "max_norm <= 0" coincide).

Under batch sharding the norm must be taken over the WHOLE batch gradient (grad.py:66-68): pass a process group and the
squared norm is all-reduced (one float over RCCL) between the two launches.  `shard_loss_scale` is the factor between the loss
this rank back-propagates and the job's objective: under DistributedDataParallel every rank differentiates the mean over ITS
shard and DDP averages the parameter gradients, so the gradient the single-process reference would have clipped is this rank's
times 1 / world_size -- pass that, and `max_norm` (a checkpointed buffer) and the clipping decision are those of the reference on
the concatenated batch whatever the number of GPUs; 1.0 (default) if the caller already scales its loss by local_B / global_B.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

_SQNORM_BLOCKS = 512  # LC_SQNORM_BLOCKS (include/lc_amd.h)


def _launch_sqnorm(grads, workspace, state):
    """Sum of squares over all tensors of the hook call -> float32 device scalar; also snapshots `state` (the running
    maximum before this call) into the workspace so that `_launch_apply` may overwrite `state` in place."""
    lib = _lib.load()
    dev = grads[0].device
    partials, ticket, snap = workspace
    sq = torch.empty((), device=dev, dtype=torch.float32)
    with _lib.on_device(dev):
        for i, g in enumerate(grads):
            last = i == len(grads) - 1
            rc = lib.lc_sqnorm(_lib.ptr(g), _lib.MAP_DTYPES[g.dtype], g.numel(), _lib.ptr(partials), _lib.ptr(ticket), _lib.ptr(sq), int(i > 0),
                               _lib.ptr(state) if last else None, _lib.ptr(snap) if last else None, _lib.stream_ptr(dev))
            _lib.check(rc, "lc_sqnorm")
    return sq, snap


def _launch_apply(grads, sq, state_before, state, initial_max_norm, scale, momentum):
    """-> (clipped tensors, total norm).  `state` (float32 device scalar) is updated IN PLACE from `state_before` (its
    snapshot): fixed addresses, so the two launches can be replayed inside a hipGraph; the update rides on the last tensor."""
    lib = _lib.load()
    dev = grads[0].device
    norm = torch.empty((), device=dev, dtype=torch.float32)
    outs = []
    with _lib.on_device(dev):
        for i, g in enumerate(grads):
            o = torch.empty_like(g)
            last = i == len(grads) - 1
            rc = lib.lc_norm_clip_apply(_lib.ptr(g), _lib.MAP_DTYPES[g.dtype], g.numel(), _lib.ptr(sq), _lib.ptr(state_before), float(initial_max_norm),
                                        float(scale), float(momentum), _lib.ptr(o), _lib.ptr(state) if last else None,
                                        _lib.ptr(norm) if last else None, _lib.stream_ptr(dev))
            _lib.check(rc, "lc_norm_clip_apply")
            outs.append(o)
    return outs, norm


class NormClipper(torch.nn.Module):
    def __init__(self, initial_max_norm=100, rel_thresh=0.7, momentum=0.1, group=None, shard_loss_scale=1.0) -> None:
        super().__init__()
        self.initial_max_norm = initial_max_norm
        self.register_buffer("max_norm", torch.tensor(-1, dtype=torch.float))
        self.momentum = momentum
        self.scale = 1 + rel_thresh
        self.last_norm = 0
        self.group = group
        self.shard_loss_scale = float(shard_loss_scale)
        self._workspace = {}

    def _ws(self, dev):
        if dev not in self._workspace:
            self._workspace[dev] = (torch.empty(_SQNORM_BLOCKS, device=dev, dtype=torch.float64),
                                    torch.zeros(1, device=dev, dtype=torch.int32), torch.empty((), device=dev, dtype=torch.float32))
        return self._workspace[dev]

    def forward(self, grads, norm_type=2):
        return self.clip(grads, norm_type)

    @torch.no_grad()
    def clip(self, grads, norm_type=2):
        if float(norm_type) != 2.0:
            raise NotImplementedError("lc_amd NormClipper: only the 2-norm the reference uses is implemented")
        single = isinstance(grads, Tensor)
        tensors = [grads] if single else list(grads)
        if not tensors:
            return tensors
        # fp16 / bf16 gradients of a mixed-precision head are read and written in their own type (fp32 arithmetic inside): no cast each way,
        # and the hook returns the dtype it was given (autograd rejects a hook that changes it)
        tensors = [_lib.require_hip_map("grad", g) for g in tensors]
        dev = tensors[0].device
        if self.max_norm.device != dev:  # module left on the CPU: the state follows the gradients (once)
            self.max_norm = self.max_norm.to(device=dev)
        sq, before = _launch_sqnorm(tensors, self._ws(dev), self.max_norm)
        if self.group is not None:
            import torch.distributed as dist

            dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            if self.shard_loss_scale != 1.0:
                sq.mul_(self.shard_loss_scale * self.shard_loss_scale)  # the norm of the job's gradient, not of this rank's loss
        clipped, self.last_norm = _launch_apply(tensors, sq, before, self.max_norm, self.initial_max_norm, self.scale, self.momentum)
        return clipped[0] if single else clipped
