"""RANSAC initialiser: single launch (one workgroup per pose) vs the split form (hypotheses / scoring / selection), event-timed.

    python scripts/ubench/ransac_forms.py            # one line per (B, N) and form
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.pnp import gpu_solver  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(best)[len(best) // 2]


def main():
    dev = torch.device("cuda:0")
    for (B, N) in ((64, 1024), (64, 2048), (256, 1024), (256, 64), (16, 1024), (1, 1024)):
        bt = synth.make_batch(B, N, seed=2, outlier_frac=0.2)
        K, X, U = bt["K"].to(dev), bt["pts3d"].to(dev), bt["pts2d"].to(dev)
        row = []
        for split in (False, True):
            us = timed(lambda: gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, split=split))
            # the same call captured in a hipGraph: kernel time without the Python/launch overhead
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, split=split)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, split=split)
            gus = timed(g.replay)
            row.append((us, gus))
        print(f"B={B:4d} N={N:5d}  single launch {row[0][0]:7.1f} us (graph {row[0][1]:7.1f})   split {row[1][0]:7.1f} us (graph {row[1][1]:7.1f})")


if __name__ == "__main__":
    main()
