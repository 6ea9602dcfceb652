#!/usr/bin/env python3
"""The sparse head's test-time chain (test.py:47-64 at configs/gsplmo.yaml: 16 keypoints, solvers ransac + weighted) on 64 objects: eager and replayed;
under rocprofv3 --kernel-trace the per-kernel table (scripts/ubench/kernel_times.sh)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.inference import GraphedSolvePnP, solve_pnp  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))
b = synth.make_batch(B, 16, seed=3, noise_px=0.3, outlier_frac=0.0)
gt = {k: v.to(dev) for k, v in dict(out_K=b["K"], pts3d=b["pts3d"], pose_best=b["pose"]).items()}
net = {k: v.to(dev) for k, v in dict(pts2d=b["pts2d"], pts2d_std=1 / b["inv_std"]).items()}
cfg = AttrDict(rel_reproj_err=False, solvers=["ransac", "weighted"])


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


eager = solve_pnp(cfg, net, gt)
solver = GraphedSolvePnP(cfg, net, gt)
solver.graph.replay()
torch.cuda.synchronize()
print(f"sparse chain, {B} objects x 16 keypoints: replay equals eager: {all(torch.equal(solver._res[k], eager[k]) for k in eager)}")
print(f"eager  {timeit(lambda: solve_pnp(cfg, net, gt)):7.1f} us per call")
print(f"replay {sorted(timeit(solver.graph.replay) for _ in range(7))[3]:7.1f} us per call")
