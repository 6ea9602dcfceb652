#!/usr/bin/env python3
"""A/B of the split solve (several workgroups per pose, lc_pnp_lm3_f32 + workspace) against one workgroup per pose at test-time shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.pnp import pnp_ceres  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


for B, N, used in ((64, 16384, 3300), (64, 16384, 2700), (64, 4096, 4096), (32, 4096, 3000), (128, 4096, 3000), (64, 16384, 12000), (16, 16384, 16384)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1, outlier_frac=0.0, noise_px=0.7).items()}
    counts = torch.full((B,), used, dtype=torch.int32, device=dev)
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts)
    g = {}
    for split in (False, True):
        graph = torch.cuda.CUDAGraph()
        ws = torch.zeros(pnp_ceres.SPLIT_WORKSPACE_MAX_BYTES, device=dev, dtype=torch.uint8)
        pnp_ceres.solve_device(*args, split=split)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph), pnp_ceres.owned_split_workspace(ws):
            out = pnp_ceres.solve_device(*args, split=split, return_iters=True)
        g[split] = (timed(graph.replay), out, ws)
    it = g[True][1][3].float().mean().item()
    same = torch.equal(g[True][1][3], g[False][1][3])
    print(f"B {B:4d} N {N:6d} used {used:6d}: one workgroup per pose {g[False][0]:7.2f} us, split {g[True][0]:7.2f} us  (LM iterations {it:.2f}, same schedule {same}, "
          f"max |dpose| {float((g[True][1][0] - g[False][1][0]).abs().max()):.2e})")
