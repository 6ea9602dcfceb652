#!/usr/bin/env python3
"""RANSAC split form: three launches (init3) vs the ticketed two launches (init4), with and without the fused inlier re-selection;
each variant replayed as a hipGraph (launch overhead out of the picture).  Shape of the test-time pipeline: 64 objects, 1024
padded correspondences, ~half of them selected."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.dense import dense_select  # noqa: E402
from lc_amd.pnp import gpu_solver  # noqa: E402

dev = torch.device("cuda:0")
B, N = int(os.environ.get("B", 64)), int(os.environ.get("N", 1024))
b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1, outlier_frac=0.3, noise_px=0.7).items()}
g = torch.Generator().manual_seed(0)
counts = torch.randint(int(N * float(os.environ.get("FILL", 0.3))), int(N * float(os.environ.get("FILL_HI", 0.55))) + 1, (B,), generator=g).to(torch.int32).to(dev)
w = torch.rand(B, N, 2, generator=g).to(dev) + 0.1


def run(ticketed, select):
    sel = dict(weights=w, min_count=4) if select == "fused" else None
    st, inl, bad = gpu_solver.solve_device(b["K"], b["pts3d"], b["pts2d"], counts, reprojectionError=3.0, refine=False, split=True,
                                           ticketed=ticketed, select=sel)
    if select == "separate":
        dense_select(b["pts2d"], w, b["pts3d"], "mask", mask=inl, counts=counts, square_weights=False, min_count=4)
    return st


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


print(f"B={B} N={N} counts {int(counts.min())}..{int(counts.max())}")
for name, t, s in (("three launches", False, None), ("three launches + dense_select", False, "separate"),
                   ("three launches, fused selection", False, "fused"), ("ticketed", True, None),
                   ("ticketed + fused selection", True, "fused")):
    print(f"{name:32s} {replay_us(lambda: run(t, s)):7.2f} us per replay")
