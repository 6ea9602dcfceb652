#!/usr/bin/env python3
"""Where does a PnP wave spend its cycles?  Diagnostic build (-DLC_TRACE_CLOCK: the trace kernel's last column carries the shader
clock since the wave started) at the metric shape: cycles up to the end of iteration 1 (loads, quaternion -> angle-axis, the
initial evaluation, the first LM iteration) and per later iteration.  Never quote this build's run time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

so = build.build_variant("traceclock", ["-DLC_TRACE_CLOCK"])
os.environ["LC_AMD_LIB"] = so
from lc_amd import synth  # noqa: E402
from lc_amd.pnp import pnp_ceres  # noqa: E402

dev = torch.device("cuda:0")
B, N = 256, 64
b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=0).items()}
for rep in range(3):
    st, tr, ret, it, trace = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], return_iters=True, trace_rows=50)
    torch.cuda.synchronize()
it = it.cpu().numpy()
clk = trace.cpu().numpy()[:, :, 7]
first = clk[:, 0]
print(f"iterations: mean {it.mean():.2f} max {it.max()}")
print(f"cycles to the end of iteration 1 (loads + set-up + initial evaluation + first iteration): median {np.median(first):.0f} (p10 {np.quantile(first, .1):.0f}, p90 {np.quantile(first, .9):.0f})")
for k in range(1, int(it.max())):
    m = it > k
    d = clk[m, k] - clk[m, k - 1]
    print(f"iteration {k + 1}: {int(m.sum()):3d} poses, median {np.median(d):.0f} cycles (p10 {np.quantile(d, .1):.0f}, p90 {np.quantile(d, .9):.0f})")
tot = clk[np.arange(B), it - 1]
print(f"whole solve: median {np.median(tot):.0f}, max {tot.max():.0f} cycles")
