#!/usr/bin/env python3
"""Which GPU kernels make up one dense `Loss_fn` step (forward + backward)?  torch profiler, kernel names with counts and device time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from tests.golden.gen_golden_lossfn import BIN_CFG, DENSE_CFG, SPARSE_CFG, bin_inputs, dense_inputs, sparse_inputs  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda:0")
for kind, cfg, make in (("dense B=32 64x64", DENSE_CFG, lambda: dense_inputs(B=32, H=64, W=64)), ("sparse B=256 N=64", SPARSE_CFG, lambda: sparse_inputs(B=256, N=64)),
                        ("bin B=32 64x64", BIN_CFG, lambda: bin_inputs(B=32, H=64, W=64)), ("bin B=32 128x128", BIN_CFG, lambda: bin_inputs(B=32, H=128, W=128))):
    fn = Loss_fn(AttrDict(cfg), AttrDict(), 17 if kind.startswith("bin") else 0).to(dev)
    gt, out = make()
    gt = {k: (v.to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}  # loader tensors are contiguous (the generator hands out strided views)
    out = {k: v.to(dev).contiguous() for k, v in out.items()}
    if len(sys.argv) > 1 and sys.argv[1] not in kind:
        continue

    def step(i):
        np.random.seed(i)
        leaves = {k: v.detach().requires_grad_(True) for k, v in out.items()}
        ld, wd = fn(gt, leaves, 1, 1000 + i, 10)
        torch.autograd.grad(sum(wd.values()), list(leaves.values()), allow_unused=True)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
    rows = [(e.key, e.count / 10, e.device_time_total / 10) for e in prof.key_averages() if e.device_time_total > 0]
    rows.sort(key=lambda r: -r[2])
    print(f"## {kind}: {sum(r[1] for r in rows):.0f} kernels, {sum(r[2] for r in rows):.0f} us of GPU time per step")
    for k, c, t in rows[:40]:
        print(f"  {c:5.1f} x {t / max(c, 1e-9):7.2f} us  = {t:7.1f}  {k[:130]}")
