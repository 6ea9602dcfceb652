#!/usr/bin/env python3
"""Host-side cost of the Python call surface around one fused launch (B=256, N=64): raw C-ABI call, `_launch_loss`
(allocations + call), the autograd Function forward, forward + backward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import _lib, synth  # noqa: E402
from lc_amd.cov_mixed import Loss_cov_mixed, _launch_loss  # noqa: E402

dev = torch.device("cuda:0")
b = {k: v.to(dev) for k, v in synth.make_batch(256, 64, seed=0).items()}
lib = _lib.load()
P = _lib.ptr
loss = torch.empty(256, device=dev)
du, ds, dx = torch.empty_like(b["pts2d"]), torch.empty_like(b["inv_std"]), torch.empty_like(b["pts3d"])


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def raw():
    lib.lc_cov_loss3_fwd_bwd_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]), None, 256, 64, 32.0, 3.0, 4.0, 0, P(loss), P(du), P(ds), P(dx), None, None, 0, _lib.stream_ptr(dev))


def launch():
    _launch_loss(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], None, 32.0, 3.0, 4.0, True, True)


u = b["pts2d"].clone().requires_grad_(True)
s = b["inv_std"].clone().requires_grad_(True)


def fwd():
    return Loss_cov_mixed(b["K"], b["pose"], b["pts3d"], u, s, None, bbox_3d=b["bbox_3d"])


def fwd_bwd():
    torch.autograd.grad(fwd().mean(), (u, s))


print(f"raw C-ABI call            {timeit(raw):7.1f} us per call (kernel itself ~8.5 us)")
print(f"_launch_loss              {timeit(launch):7.1f} us")
print(f"Loss_cov_mixed forward    {timeit(fwd):7.1f} us")
print(f"forward + mean + backward {timeit(fwd_bwd, 1000):7.1f} us")
