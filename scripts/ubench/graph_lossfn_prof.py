#!/usr/bin/env python3
"""What a replayed sparse `Loss_fn` step (GraphedLoss) consists of: torch profiler table of 20 steps (host ops and GPU kernels)."""
import os
import sys
import warnings

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.graphs import GraphedLoss  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from tests.golden.gen_golden_lossfn import DENSE_CFG, SPARSE_CFG, dense_inputs, sparse_inputs  # noqa: E402

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "sparse"
cfg, make = (SPARSE_CFG, lambda: sparse_inputs(B=256, N=64)) if kind == "sparse" else (DENSE_CFG, lambda: dense_inputs(B=32, H=64, W=64))
gt, out = make()
gt = {k: (v.to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
out = {k: v.to(dev).contiguous() for k, v in out.items()}
graphed = GraphedLoss(Loss_fn(AttrDict(cfg), AttrDict(), 0).to(dev), gt, out, 1, 10_000, 10)


def step():
    leaves = {k: v.detach().requires_grad_(True) for k, v in out.items()}
    ld, wd = graphed(gt, leaves)
    torch.autograd.grad(sum(wd.values()), list(leaves.values()), allow_unused=True)


np.random.seed(0)
for _ in range(30):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=70))
