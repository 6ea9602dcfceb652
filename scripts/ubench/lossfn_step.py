#!/usr/bin/env python3
"""Wall time of a whole `Loss_fn` step (forward + backward to the network outputs) on the GPU, sparse and dense configs:
how much of it is the HIP kernels and how much is host-side torch glue (launch-bound small ops)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from tests.golden.gen_golden_lossfn import SPARSE_CFG, DENSE_CFG, BIN_CFG, sparse_inputs, dense_inputs, bin_inputs  # noqa: E402


def run(kind, make, cfg, reps=50):
    dev = torch.device("cuda:0")
    fn = Loss_fn(AttrDict(cfg), AttrDict(), 17 if kind.startswith("bin") else 0).to(dev)
    gt, out = make()
    gt = {k: (v.to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}  # loader tensors are contiguous (the generator hands out strided views)
    out = {k: v.to(dev).contiguous() for k, v in out.items()}

    def step(i):
        np.random.seed(i)
        leaves = {k: v.detach().requires_grad_(True) for k, v in out.items()}
        ld, wd = fn(gt, leaves, 1, 1000 + i, 10)
        total = sum(wd.values())
        torch.autograd.grad(total, list(leaves.values()), allow_unused=True)

    for i in range(5):
        step(i)
    win = []
    for _ in range(5):  # median of 5 windows (the shared pool shows occasional ~55 ms stalls)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            step(i)
        torch.cuda.synchronize()
        win.append((time.perf_counter() - t0) / reps)
    dt = sorted(win)[2]
    print(f"{kind}: {dt * 1e3:.3f} ms per Loss_fn step (forward + backward)")


if __name__ == "__main__":
    run("sparse B=256 N=64", lambda: sparse_inputs(B=256, N=64), SPARSE_CFG)
    run("dense B=32 64x64", lambda: dense_inputs(B=32, H=64, W=64), DENSE_CFG)
    run("bin B=32 64x64", lambda: bin_inputs(B=32, H=64, W=64), BIN_CFG)
