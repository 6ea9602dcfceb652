"""Sharding of the per-object batch over the GPUs of one node (one process per GPU, `torch.distributed`; backend "nccl"
is RCCL on ROCm, "gloo" on CPU for tests).

The hot path itself needs NO collective: every pose is independent, so each rank runs the fused kernels on
its contiguous slice of the batch.  What a training step needs around it:
  * `allreduce_gradients`  -- mean of parameter gradients across ranks in few large flat buckets (xGMI is point-to-point,
    7 links x ~153 GB/s: ring collectives are per-link bound, so prefer a handful of 64 MiB buckets over many small ones);
  * `global_mean`          -- the batch-mean loss for logging (`losses.py:334,386` take `.mean()` over the whole batch);
  * `NormClipper(group=)`  -- whole-batch gradient norm (one float) in `lc_amd/grad.py`.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"))
    return rank, world, device


def shard_range(n: int, rank: int, world: int):
    """Contiguous [lo, hi) of rank's share of n items (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    n = next(iter(batch.values())).shape[0]
    lo, hi = shard_range(n, rank, world)
    return {k: (v[lo:hi] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}


def global_mean(local_sum: torch.Tensor, local_count: int, group=None) -> torch.Tensor:
    """Mean over the whole (sharded) batch of a per-sample quantity given this rank's sum and count."""
    buf = torch.stack((local_sum.detach().to(torch.float64).reshape(()), torch.tensor(float(local_count), dtype=torch.float64,
                                                                                     device=local_sum.device)))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[0] / buf[1]).to(local_sum.dtype)


def aggregate_regions(region_seconds, steps: int, device=None, group=None) -> Dict[str, object]:
    """bench.py's timing aggregation: every rank timed the same R regions of `steps` steps (each bracketed by a barrier);
    a region lasts as long as its slowest rank (MAX over ranks), the reported step time is the MEDIAN region.  Also returns
    what proves the collective saw every rank: world size, backend, per-rank median ms/step."""
    t = torch.tensor(list(region_seconds), dtype=torch.float64, device=device or "cpu")
    world, backend = 1, "none"
    per_rank = t[None]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        world, backend = dist.get_world_size(group), dist.get_backend(group)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t, group=group)
        per_rank = torch.stack(bufs)
    per_rank = per_rank.cpu()
    region_max = per_rank.max(0).values
    srt = region_max.sort().values
    R = len(srt)
    q = lambda f: float(srt[min(R - 1, int(f * R))])  # noqa: E731
    med = float(region_max.median())
    return {"ranks_seen": world, "backend": backend, "regions": R, "median_region_s": med, "ms_per_step": med / steps * 1e3,
            "region_ms_per_step": {"min": q(0.0) / steps * 1e3, "p10": q(0.1) / steps * 1e3, "p50": med / steps * 1e3,
                                   "p90": q(0.9) / steps * 1e3, "max": float(srt[-1]) / steps * 1e3},
            "per_rank_ms_per_step": [float(r.median()) / steps * 1e3 for r in per_rank]}


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20, average: bool = True):
    """Sum (or average) `.grad` across ranks through flat buckets of ~bucket_bytes.  TEST HELPER (blocking, after backward):
    the world-size-2 gloo tests use it to check the sharded step against the full batch; training scripts wrap the model in
    `torch.nn.parallel.DistributedDataParallel` instead (examples/train_*_ddp.py), whose bucketed all-reduce overlaps backward."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for g in grads:
        if bucket and (bucket[0].dtype != g.dtype or size + g.numel() * g.element_size() > bucket_bytes):
            flush()
        bucket.append(g)
        size += g.numel() * g.element_size()
    flush()
