#!/usr/bin/env python3
"""Does a PnP wave slow down in CYCLES or in TIME when the whole chip is busy?  Diagnostic build (-DLC_TRACE_CLOCK: shader-clock count
of every wave's whole solve, trace kernel) against the event-timed duration of the same launch, at B = 256 (one wave on a quarter of
the SIMDs), 512, 1024 (one wave on every SIMD) and 2048 (two).  cycles / time = the clock the launch ran at."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

os.environ["LC_AMD_LIB"] = build.build_variant("traceclock", ["-DLC_TRACE_CLOCK"])
from lc_amd import synth  # noqa: E402
from lc_amd.pnp import pnp_ceres  # noqa: E402

dev = torch.device("cuda:0")
for B in (256, 512, 1024, 2048):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, 64, seed=0).items()}
    run = lambda: pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], return_iters=True, trace_rows=50)  # noqa: E731
    for _ in range(3):
        st, tr, ret, it, trace = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st, tr, ret, it, trace = run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    it = it.cpu().numpy()
    clk = trace.cpu().numpy()[:, :, 7]
    tot = clk[np.arange(B), it - 1]
    print(f"B={B:5d}: trace-kernel launch {us:6.1f} us (host-inclusive); slowest wave {tot.max():.0f} cycles, median wave {np.median(tot):.0f} cycles"
          f" -> >= {tot.max() / us / 1e3:.2f} GHz if the launch lasted as long as its slowest wave", flush=True)
