#!/usr/bin/env python3
"""Soak of the dense pose unit (lc_pose_unit2_f32, one launch) against the two stand-alone launches: many launches of several shapes with
other work in between, every output compared bit for bit; reports WHICH output differed if any does.  usage: dense_unit_soak.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import synth  # noqa: E402
from lc_amd.cov_mixed import loss_cov_mixed_fused  # noqa: E402
from lc_amd.fused import PoseUnit  # noqa: E402
from lc_amd.pnp import pnp_ceres  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
cases = []
for B, N in ((32, 1849), (32, 1024), (5, 700), (64, 1024), (3, 2048)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N).items()}
    go = torch.rand(B, device=dev) + 0.5
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
    st, tr, ret, it = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], return_iters=True)
    cases.append((B, N, b, go, dict(loss=loss, d_pts2d=du, d_inv_std=ds, d_pts3d=dx, states=st, trust_radius=tr, invalid=ret, iters=it), PoseUnit(B, N, dev)))
torch.cuda.synchronize()
junk = []
t0, launches, bad = time.time(), 0, {}
while time.time() - t0 < budget:
    for B, N, b, go, want, unit in cases:
        for t in (unit.loss, unit.d_pts2d, unit.d_inv_std, unit.d_pts3d, unit.states):
            t.fill_(float("nan"))
        unit(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"], grad_out=go)
        junk.append(torch.randn(1 << 16, device=dev))  # allocator churn between the launches
        if len(junk) > 64:
            junk.clear()
        torch.cuda.synchronize()
        launches += 1
        for k, w in want.items():
            if not torch.equal(getattr(unit, k), w):
                bad[(B, N, k)] = bad.get((B, N, k), 0) + 1
        # the stand-alone launches again, against their own first results
        loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
        st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
        for k, (x, w) in dict(loss=(loss, want["loss"]), d_pts2d=(du, want["d_pts2d"]), states=(st, want["states"])).items():
            if not torch.equal(x, w):
                bad[(B, N, "standalone " + k)] = bad.get((B, N, "standalone " + k), 0) + 1
print(f"dense pose unit soak: {launches} one-launch units over 5 shapes (and as many stand-alone pairs) compared bit for bit, mismatches: {bad if bad else 0}, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
