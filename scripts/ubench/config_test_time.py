#!/usr/bin/env python3
"""The test-time pipeline at a reference config's own knobs (CFG=zlmo | glmo, B objects; lc_amd.synth.TEST_TIME_CONFIGS): eager call and hipGraph
replay, results compared.  Under `rocprofv3 --kernel-trace` this is the trace the per-kernel table of profiles/r04/test_time/ comes from
(scripts/ubench/kernel_avgs.py averages the last launches of every kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.inference import GraphedSolvePnP, solve_pnp  # noqa: E402

dev = torch.device("cuda:0")
name, B = os.environ.get("CFG", "zlmo"), int(os.environ.get("B", 64))
cfg, gt, out = synth.test_time_inputs(name, B=B, seed=3)
cfg = AttrDict(cfg)
gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
out = {k: v.to(dev).contiguous() for k, v in out.items()}
if os.environ.get("MAPS") == "bf16":  # the maps a bf16-autocast backbone hands over, read natively
    out = {k: (v.to(torch.bfloat16) if v.is_floating_point() and k != "xyz_weights_scale" else v) for k, v in out.items()}


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


eager = solve_pnp(cfg, out, gt)
solver = GraphedSolvePnP(cfg, out, gt)
solver.graph.replay()
torch.cuda.synchronize()
print(f"{name}: {B} objects, {cfg}")
print(f"captured: results equal to eager: {all(torch.equal(solver._res[k], eager[k]) for k in eager)}")
print(f"eager  {timeit(lambda: solve_pnp(cfg, out, gt)):7.1f} us per call")
runs = sorted(timeit(solver.graph.replay, 200) for _ in range(9))
print(f"replay {runs[4]:7.1f} us per call (median of 9 x 200 replays; min {runs[0]:.1f}, max {runs[-1]:.1f})")
