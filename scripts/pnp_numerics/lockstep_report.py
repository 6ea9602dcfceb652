#!/usr/bin/env python3
"""GPU box: per problem set, how far the HIP kernel's trust-region schedule is from the oracle's, row by row
(the numbers behind tests/test_gpu_pnp_trace.py).   python scripts/pnp_numerics/lockstep_report.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.pnp_cases import PNP_CASES, pnp_case  # noqa: E402
from tests.test_gpu_pnp_trace import lockstep_mask, run_both  # noqa: E402

NAMES = ["kind", "cost", "cand_cost", "model_cost_change", "rho", "step_norm", "radius", "gmax"]
for name in PNP_CASES:
    c = pnp_case(name)
    (st, tr, ret, it, trace), (so, tro, reto, ito, traceo) = run_both(c)
    same = lockstep_mask(it, trace, ito, traceo)
    line = f"{name}: jobs {len(ret)} lock-step {int(same.sum())} flag flips {int((ret != reto).sum())} iters max {int(it.max())} |"
    k, o = trace[same], traceo[same]
    for col in range(1, 8):
        a, b = k[:, :, col], o[:, :, col]
        m = np.isfinite(b) & (np.abs(b) < 1e300) & (b != 0)
        if col == 4:
            m &= np.abs(b) < 10
        rel = np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300) if m.any() else np.zeros(1)
        line += f" {NAMES[col]} {rel.max():.1e}"
    print(line, flush=True)
    for j in np.nonzero(~same)[0][:6]:
        d = trace[j, :, 0] != traceo[j, :, 0]
        split = int(np.argmax(d)) if d.any() else min(it[j], ito[j])
        r = max(split - 1, 0)
        print(f"   job {j}: iters {it[j]}/{ito[j]} ret {ret[j]}/{reto[j]} split at row {split}: kernel {trace[j, split, :5]} oracle {traceo[j, split, :5]}"
              f" | row before: rho {trace[j, r, 4]:.6g}/{traceo[j, r, 4]:.6g} radius {trace[j, r, 6]:.6g}/{traceo[j, r, 6]:.6g}", flush=True)
