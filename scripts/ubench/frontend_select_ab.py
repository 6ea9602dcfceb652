#!/usr/bin/env python3
"""Front end + point selection: the two launches against the one launch (lc_dense_frontend_select3), replayed as hipGraphs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.dense import dense_front_end_select, dense_front_end_with_visibility, dense_select  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def replay_us(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            fn()
        for _ in range(10):
            graph.replay()
        out = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(out)[3]


for B, H, W, sample in ((64, 64, 64, 2), (32, 86, 86, 2), (64, 128, 128, 2), (16, 128, 128, 2), (64, 128, 128, 1)):
    N = -(-H // sample) * -(-W // sample)
    if N > 8192:
        continue
    xyz, wl = torch.randn(B, 3, H, W, generator=g).to(dev), torch.randn(B, 2, H, W, generator=g).to(dev)
    ws, ns, vl = (torch.rand(B, generator=g) + 0.5).to(dev), (torch.rand(B, 3, generator=g) + 0.5).to(dev), torch.randn(B, 1, H, W, generator=g).to(dev)

    def two():
        u, s_, x, vis = dense_front_end_with_visibility(xyz, wl, ws, ns, vl, 0.5, sample=sample)
        return dense_select(u, s_, x, "quantile_in_mask", mask=vis, quantile=0.5)

    def one():
        return dense_front_end_select(xyz, wl, ws, ns, vl, "quantile_in_mask", quantile=0.5, sample=sample)
    print(f"B={B:3d} {H}x{W} stride {sample} (N={N:5d}): two launches {replay_us(two):6.2f} us   one launch {replay_us(one):6.2f} us")
