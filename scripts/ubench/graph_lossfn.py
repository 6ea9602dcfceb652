#!/usr/bin/env python3
"""`Loss_fn` step (forward + backward) eager vs replayed as hipGraphs (`lc_amd.graphs.GraphedLoss`), sparse and dense heads."""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.graphs import GraphedLoss  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from tests.golden.gen_golden_lossfn import DENSE_CFG, SPARSE_CFG, dense_inputs, sparse_inputs  # noqa: E402

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")


def timeit(f, n=100):
    for _ in range(20):
        f()
    out = []
    for _ in range(7):  # median of 7 windows: the shared pool shows occasional ~55 ms stalls
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(out)[3]


for kind, cfg, make in (("sparse B=256 N=64", SPARSE_CFG, lambda: sparse_inputs(B=256, N=64)),
                        ("dense B=32 64x64", DENSE_CFG, lambda: dense_inputs(B=32, H=64, W=64))):
    gt, out = make()
    gt = {k: (v.to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}  # loader tensors are contiguous (the generator hands out strided views)
    out = {k: v.to(dev).contiguous() for k, v in out.items()}
    fn = Loss_fn(AttrDict(cfg), AttrDict(), 0).to(dev)
    graphed = GraphedLoss(Loss_fn(AttrDict(cfg), AttrDict(), 0).to(dev), gt, out, 1, 10_000, 10)

    def step(call):
        leaves = {k: v.detach().requires_grad_(True) for k, v in out.items()}
        ld, wd = call(gt, leaves)
        torch.autograd.grad(sum(wd.values()), list(leaves.values()), allow_unused=True)

    np.random.seed(0)
    te = timeit(lambda: step(lambda g, o: fn(g, o, 1, 10_000, 10)))
    np.random.seed(0)
    tg = timeit(lambda: step(graphed))
    print(f"{kind}: eager {te:7.1f} us  graphed {tg:7.1f} us per step (forward + backward)")
