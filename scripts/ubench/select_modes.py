#!/usr/bin/env python3
"""lc_dense_select_kernel per mode, replayed as a hipGraph (B objects x N candidates; half of the mask set)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.dense import dense_select  # noqa: E402

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
        for _ in range(7):  # median of 7 windows: the shared pool shows occasional ~55 ms stalls
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(out)[3]


for B, N in ((64, 1024), (64, 4096), (256, 1024)):
    u, w, x = torch.rand(B, N, 2, generator=g).to(dev), (torch.rand(B, N, 2, generator=g) + 0.05).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
    m = (torch.rand(B, N, generator=g) < 0.5).to(dev)
    row = [f"B={B:4d} N={N:5d}"]
    for mode in ("mask", "quantile", "quantile_in_mask"):
        row.append(f"{mode} {replay_us(lambda: dense_select(u, w, x, mode, mask=m, quantile=0.5)):6.2f} us")
    print("   ".join(row))
