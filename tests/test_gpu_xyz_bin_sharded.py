"""`Loss_xyz_bin` (losses.py:196-216) when the batch is sharded over ranks: the counts-out / weights-in form of the one-launch kernel
(`lc_xyz_bin_loss_counts` -> int64 all-reduce -> `lc_xyz_bin_loss_finish`, include/lc_amd.h).

  (a) a process group of ONE rank: loss, histogram trajectory, bit weights and gradient are the one-launch kernel's, bit for bit, for fp32 /
      fp16 / bf16 maps -- at the reference's own training shape (the `lossfn_bin_zlmo` fixture: 128x128 maps, 21 code planes) the split form
      therefore inherits the one-launch kernel's parity with the reference's `Loss_fn`;
  (b) two ranks sharing the one GPU (gloo), half of the batch each, three steps, 16-bit maps included: the histogram on every rank is the
      single process' on the concatenated batch bit for bit (integer counts), each rank's gradient is `world` x its slice of the single
      process' gradient, the mean of the ranks' losses is the single process' loss."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
SCALE = 4096.0  # what a GradScaler does around an fp16 step: the code loss's gradient entries (~1e-5) are below fp16's normal range otherwise


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(B, C, H, W, dtype, seed=4):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(B, C, H, W, generator=g) * 2).to(dtype)
    bits = (logits > 0) ^ (torch.rand(B, C, H, W, generator=g) < 0.25)
    vis = torch.randn(B, 1, H, W, generator=g).to(dtype)
    return logits, bits, vis


@pytest.fixture(scope="module")
def one_rank_group():
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_port()}", rank=0, world_size=1)
    yield dist.group.WORLD
    dist.destroy_process_group()


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(4, 21, 128, 128), (3, 17, 16, 16), (2, 9, 15, 13), (1, 128, 8, 8), (2, 9, 6, 6)], ids=["zlmo", "small", "ragged", "C128", "hw36"])
def test_one_rank_group_is_the_one_launch_kernel(one_rank_group, shape, dt):
    from lc_amd import dense_aux

    dev = torch.device("cuda:0")
    B, C, H, W = shape
    logits, bits, vis = (t.to(dev) for t in _inputs(B, C, H, W, DTYPES[dt]))
    h_one = torch.full((C,), 0.5, device=dev)
    h_two = h_one.clone()
    for step in range(3):
        xs = []
        for hist, group in ((h_one, None), (h_two, one_rank_group)):
            x = (logits * (1 + 0.3 * step)).clone().requires_grad_(True)
            loss = dense_aux.xyz_bin_loss(x, bits, vis, hist, 0.05, group=group)
            (loss * 3).backward()
            xs.append((loss.detach().clone(), x.grad.clone()))
        assert torch.equal(xs[0][0], xs[1][0]) and torch.equal(xs[0][1], xs[1][1]) and torch.equal(h_one, h_two)
    assert float((h_one - 0.5).abs().max()) > 1e-3


def test_loss_fn_with_a_one_rank_group_on_the_reference_trajectory(one_rank_group):
    """The reference's own `Loss_fn` at zlmo's training shape (tests/golden/lossfn_bin_zlmo.npz: B=4, 128x128 maps, 21 code planes, stride 3,
    generated from the unmodified reference): with a process group the code loss runs counts -> all-reduce -> finish and the NormClipper
    all-reduces its norm; every loss, gradient and state (max_norm, the code histogram) stays within the single-process test's tolerances."""
    import re

    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import TRAIN_KINDS, run
    from tests.util import rel_err

    z = np.load(os.path.join(GOLDEN, "lossfn_bin_zlmo.npz"))
    assert list(z["steps"]) == TRAIN_KINDS["bin_zlmo"][0]
    rec = run(lambda cfg, cfg_global, bits: Loss_fn(cfg, cfg_global, bits, group=one_rank_group), "bin_zlmo", list(z["steps"]), torch.float32,
              device=torch.device("cuda:0"))
    for k in rec:
        if k == "steps":
            continue
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec[k]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), (k, float(rec[k]), float(z[k]))
        elif "_grad_" in k:
            assert np.isfinite(rec[k]).all() and rel_err(rec[k], z[k]) <= 2e-3, (k, rel_err(rec[k], z[k]))
        else:
            assert rel_err(rec[k], z[k]) <= 1e-3, (k, rec[k], z[k])


def _worker(rank, world, port, shape, dt, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lc_amd import dist as lcd
        from lc_amd.losses import Loss_xyz_bin

        dev = torch.device("cuda:0")
        B, C, H, W = shape
        logits, bits, vis = (t.to(dev) for t in _inputs(B, C, H, W, DTYPES[dt]))
        lo, hi = lcd.shard_range(B, rank, world)
        fn = Loss_xyz_bin(C, group=dist.group.WORLD).to(dev)
        hists, losses, grads = [], [], []
        for step in range(3):
            # (scaled as the whole batch, then sliced: torch's fp16 `tensor * python float` rounds a few elements differently for another tensor size)
            x = (logits * (1 + 0.3 * step))[lo:hi].clone().requires_grad_(True)
            loss = fn(x, bits[lo:hi], vis[lo:hi])
            (loss * SCALE).backward()
            hists.append(fn.histogram.cpu().numpy().copy())
            losses.append(float(loss.detach()))
            grads.append(x.grad.float().cpu().numpy().copy())
        ret[rank] = (hists, losses, grads, (lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_the_gpu_keep_the_single_process_histogram(world, dt):
    shape = (6, 21, 32, 32)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, dt, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    from lc_amd.losses import Loss_xyz_bin

    dev = torch.device("cuda:0")
    B, C, H, W = shape
    logits, bits, vis = (t.to(dev) for t in _inputs(B, C, H, W, DTYPES[dt]))
    fn = Loss_xyz_bin(C).to(dev)
    for step in range(3):
        x = (logits * (1 + 0.3 * step)).clone().requires_grad_(True)
        loss = fn(x, bits, vis)
        (loss * SCALE).backward()
        g = x.grad.float().cpu().numpy()
        for r in range(world):
            hists, losses, grads, (lo, hi) = ret[r]
            assert np.array_equal(hists[step], fn.histogram.cpu().numpy())  # integer counts all-reduced: the same float operations on the same integers
            # each rank's mean runs over its own pixels: world x the single process' slice (up to the rounding of the gradient in the map's type)
            tol = {"f32": 2e-6, "f16": 2e-3, "bf16": 1.6e-2}[dt]
            # (atol: an entry below fp16's normal range is written in steps of 6e-8)
            np.testing.assert_allclose(grads[step] / world, g[lo:hi], rtol=tol, atol=1e-12 if dt == "f32" else 1.2e-7)
        assert abs(sum(ret[r][1][step] for r in range(world)) / world - float(loss.detach())) <= 2e-6 * max(1.0, abs(float(loss.detach())))
    assert float(np.abs(ret[0][0][-1] - 0.5).max()) > 1e-3
