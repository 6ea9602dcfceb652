#!/usr/bin/env python3
"""Can the sparse `Loss_fn` step (forward + backward) be replayed as hipGraphs (torch.cuda.make_graphed_callables)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from tests.golden.gen_golden_lossfn import SPARSE_CFG, sparse_inputs  # noqa: E402

dev = torch.device("cuda:0")
fn = Loss_fn(AttrDict(SPARSE_CFG), AttrDict(), 0).to(dev)
gt, out = sparse_inputs(B=256, N=64)
gt = {k: v.to(dev) for k, v in gt.items()}
out = {k: v.to(dev) for k, v in out.items()}
STEP = 10_000  # past the warm-up ramp: loss_pose_factor == 1


def total(pts2d, std, K, pose, pts3d, bbox):
    g = dict(gt, out_K=K, pose_best=pose, pts3d=pts3d, bbox_3d=bbox)
    ld, wd = fn(g, dict(pts2d=pts2d, pts2d_std=std), 1, STEP, 10)
    return sum(wd.values())


def args():
    return (out["pts2d"].clone().requires_grad_(True), out["pts2d_std"].clone().requires_grad_(True), gt["out_K"], gt["pose_best"],
            gt["pts3d"], gt["bbox_3d"])


def timeit(f, n=300):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def eager_step():
    a = args()
    torch.autograd.grad(total(*a), a[:2])


graphed = torch.cuda.make_graphed_callables(total, args())


def graphed_step():
    a = args()
    torch.autograd.grad(graphed(*a), a[:2])


a = args()
l0 = total(*a)
g0 = torch.autograd.grad(l0, a[:2])
b = args()
l1 = graphed(*b)
g1 = torch.autograd.grad(l1, b[:2])
print("loss equal:", torch.equal(l0, l1), " grads equal:", all(torch.equal(x, y) for x, y in zip(g0, g1)))
print(f"eager   {timeit(eager_step):7.1f} us per step")
print(f"graphed {timeit(graphed_step):7.1f} us per step")
