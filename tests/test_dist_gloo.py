"""World-size-2 gloo test of the sharded path on CPU (the kernels are replaced by the oracle via tests/cpu_backend.py):
sharding the batch over ranks + gradient all-reduce + NormClipper's whole-batch norm == the single-process full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _patch_backend(monkeypatch=None):
    """What the `oracle_backend` fixture does; without pytest's monkeypatch in the spawned children."""
    from tests import cpu_backend

    cpu_backend.apply(monkeypatch.setattr if monkeypatch is not None else setattr)


def _model_and_batch(B, N, dtype=torch.float64):
    from lc_amd import synth

    b = {k: v.to(dtype) for k, v in synth.make_batch(B, N, seed=5).items()}
    torch.manual_seed(0)
    lin = torch.nn.Linear(2, 2, dtype=dtype)  # a stand-in "network": pts2d = lin(pts2d_obs), std = softplus(head)
    head = torch.nn.Parameter(torch.zeros(2, dtype=dtype))
    with torch.no_grad():
        lin.weight.copy_(torch.eye(2) + 0.01 * torch.randn(2, 2))
        lin.bias.zero_()
    return b, lin, head


def _loss_on(b, lin, head, fn, global_B):
    gt = dict(pose_best=b["pose"], out_K=b["K"], pts3d=b["pts3d"], bbox_3d=b["bbox_3d"], msk_noc=None, msk_vis=None)
    out = dict(pts2d=lin(b["pts2d"]), pts2d_std=torch.nn.functional.softplus(head).expand_as(b["pts2d"]) + 0.5)
    loss_dict, w = fn(gt, out, 0, 100, 10)
    # Loss_fn means over the LOCAL batch; rescale so that the sum over ranks is the global-batch mean
    local_B = b["pose"].shape[0]
    return sum(w.values()) * (local_B / global_B)


def _worker(rank, world, port, B, N, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _patch_backend()
        from lc_amd import dist as lcd
        from lc_amd.config import AttrDict
        from lc_amd.grad import NormClipper
        from lc_amd.losses import Loss_fn

        b, lin, head = _model_and_batch(B, N)
        shard = lcd.shard_batch(b, rank, world)
        fn = Loss_fn(AttrDict(pose_loss_cfg=dict(clip_weight_grad=False), w_loss_kpts=1, w_loss_pose=0.7), AttrDict())
        loss = _loss_on(shard, lin, head, fn, B)
        loss.backward()
        params = list(lin.parameters()) + [head]
        lcd.allreduce_gradients(params, average=False, bucket_bytes=32)  # tiny buckets: exercises the multi-bucket path
        gm = lcd.global_mean(loss.detach() * B, shard["pose"].shape[0])  # sum of per-sample losses on this rank / count
        # whole-batch NormClipper: each rank clips its slice of one gradient tensor, norm all-reduced
        g_full = torch.arange(1, 13, dtype=torch.float64).reshape(4, 3)
        lo, hi = lcd.shard_range(4, rank, world)
        clip = NormClipper(initial_max_norm=5.0, group=dist.group.WORLD)
        clipped = clip.clip(g_full[lo:hi])
        # the DistributedDataParallel form: every rank back-propagates the mean over ITS shard (gradients world x larger than the
        # job's), shard_loss_scale = 1 / world brings norm, clipping decision and the checkpointed max_norm back to the job's
        clip_ddp = NormClipper(initial_max_norm=5.0, group=dist.group.WORLD, shard_loss_scale=1.0 / world)
        clipped_ddp = clip_ddp.clip(world * g_full[lo:hi])
        clip_ddp.clip(world * 0.5 * g_full[lo:hi])  # second call: the EMA branch
        if rank == 0:
            ret["ddp"] = (float(clip_ddp.max_norm), (clipped_ddp / world).numpy())
            ret["grads"] = [p.grad.clone().numpy() for p in params]
            ret["max_norm"] = float(clip.max_norm)
            ret["clipped0"] = clipped.numpy()
            ret["gm"] = float(gm)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_equals_full_batch(monkeypatch):
    B, N = 6, 12
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, N, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    # single process, full batch
    _patch_backend(monkeypatch)
    from lc_amd.config import AttrDict
    from lc_amd.grad import NormClipper
    from lc_amd.losses import Loss_fn

    b, lin, head = _model_and_batch(B, N)
    fn = Loss_fn(AttrDict(pose_loss_cfg=dict(clip_weight_grad=False), w_loss_kpts=1, w_loss_pose=0.7), AttrDict())
    loss = _loss_on(b, lin, head, fn, B)
    loss.backward()
    for got, p in zip(ret["grads"], list(lin.parameters()) + [head]):
        np.testing.assert_allclose(got, p.grad.numpy(), rtol=1e-9, atol=1e-12)
    g_full = torch.arange(1, 13, dtype=torch.float64).reshape(4, 3)
    clip = NormClipper(initial_max_norm=5.0)
    full_clipped = clip.clip(g_full)
    assert abs(ret["max_norm"] - float(clip.max_norm)) <= 1e-12 * float(clip.max_norm)
    np.testing.assert_allclose(ret["clipped0"], full_clipped[:2].numpy(), rtol=1e-12)
    clip.clip(0.5 * g_full)
    assert abs(ret["ddp"][0] - float(clip.max_norm)) <= 1e-12 * float(clip.max_norm)
    np.testing.assert_allclose(ret["ddp"][1], full_clipped[:2].numpy(), rtol=1e-12)


def test_shard_range_covers_everything():
    from lc_amd.dist import shard_range

    for n in (0, 1, 7, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _agg_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lc_amd import dist as lcd

        # five regions of 10 steps: rank 1 is the slow one in regions 0-2, rank 0 in regions 3-4
        regions = [[1.0, 1.1, 1.2, 3.0, 5.0], [2.0, 2.1, 2.2, 1.0, 1.0]][rank]
        agg = lcd.aggregate_regions(regions, steps=10)
        if rank == 0:
            ret["agg"] = agg
    finally:
        dist.destroy_process_group()


def test_bench_timing_aggregation_two_ranks():
    """bench.py's multi-rank timing (lc_amd.dist.aggregate_regions): a region lasts as long as its slowest rank, the median
    region is reported, and the line carries the proof that the collective saw every rank (ranks_seen, backend, per-rank times)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_agg_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    agg = ret["agg"]
    assert agg["ranks_seen"] == 2 and agg["backend"] == "gloo" and agg["regions"] == 5
    # per-region MAX over ranks = [2.0, 2.1, 2.2, 3.0, 5.0] -> median 2.2 s per 10 steps
    assert abs(agg["median_region_s"] - 2.2) < 1e-12 and abs(agg["ms_per_step"] - 220.0) < 1e-9
    assert abs(agg["region_ms_per_step"]["min"] - 200.0) < 1e-9 and abs(agg["region_ms_per_step"]["max"] - 500.0) < 1e-9
    assert [round(v, 6) for v in agg["per_rank_ms_per_step"]] == [120.0, 200.0]  # each rank's own median region
    # single process: no collective
    from lc_amd import dist as lcd

    one = lcd.aggregate_regions([0.3, 0.1, 0.2], steps=100)
    assert one["ranks_seen"] == 1 and one["backend"] == "none" and abs(one["ms_per_step"] - 2.0) < 1e-12


def _bin_inputs(B, C=9, H=12, W=10):
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(B, C, H, W, generator=g, dtype=torch.float64) * 2
    bits = (logits > 0) ^ (torch.rand(B, C, H, W, generator=g) < 0.25)
    vis = torch.randn(B, 1, H, W, generator=g, dtype=torch.float64)
    return logits, bits, vis


def _bin_worker(rank, world, port, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lc_amd import dist as lcd
        from lc_amd.losses import Loss_xyz_bin

        logits, bits, vis = _bin_inputs(B)
        lo, hi = lcd.shard_range(B, rank, world)
        fn = Loss_xyz_bin(logits.shape[1], group=dist.group.WORLD).double()
        hists, losses, grads = [], [], []
        for step in range(3):
            x = (logits[lo:hi] * (1 + 0.3 * step)).clone().requires_grad_(True)
            loss = fn(x, bits[lo:hi], vis[lo:hi])
            loss.backward()
            hists.append(fn.histogram.clone().numpy())
            losses.append(float(loss))
            grads.append(x.grad.clone().numpy())
        ret[rank] = (hists, losses, grads, (lo, hi))
    finally:
        dist.destroy_process_group()


def test_code_histogram_under_sharding_is_the_single_process_histogram():
    """SURVEY.md 8(e), collective 4: `Loss_xyz_bin.histogram` (losses.py:203-208, a checkpointed EMA of per-bit Hamming errors over the visible
    pixels of the WHOLE batch).  Two ranks with three of six samples each, three steps: the buffer on every rank equals the single process' on the
    concatenated batch, the mean of the ranks' losses is the single process' loss, and each rank's gradient x 1/world (what DDP's averaging makes
    of it) is its slice of the single process' gradient -- the bit weights come from the shared histogram."""
    B, world = 6, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_bin_worker, args=(r, world, port, B, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    from lc_amd.losses import Loss_xyz_bin

    logits, bits, vis = _bin_inputs(B)
    fn = Loss_xyz_bin(logits.shape[1]).double()
    for step in range(3):
        x = (logits * (1 + 0.3 * step)).clone().requires_grad_(True)
        loss = fn(x, bits, vis)
        loss.backward()
        for r in range(world):
            hists, losses, grads, (lo, hi) = ret[r]
            np.testing.assert_allclose(hists[step], fn.histogram.numpy(), rtol=0, atol=1e-12)
            np.testing.assert_allclose(grads[step] / world, x.grad[lo:hi].numpy(), rtol=1e-9, atol=1e-15)
        # (torch's BCE on float64 logits against float32 targets is itself only good to ~2e-8 between a batch and its halves: the reference formula's
        # own arithmetic, measured here without any sharding)
        assert abs(sum(ret[r][1][step] for r in range(world)) / world - float(loss)) <= 1e-7
    assert float(np.abs(ret[0][0][-1] - 0.5).max()) > 1e-3  # the EMA moved
