#!/usr/bin/env python3
"""LM iteration counts and event-timed launch durations of the two solves of the test-time chain at zlmo's knobs (64 objects, 16 384 candidates)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import floatbits, synth  # noqa: E402
from lc_amd.dense import dense_front_end_select  # noqa: E402
from lc_amd.pnp import gpu_solver, pnp_ceres  # noqa: E402

dev = torch.device("cuda:0")
cfg, gt, out = synth.test_time_inputs("zlmo", B=64, seed=3)
gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
out = {k: v.to(dev).contiguous() for k, v in out.items()}
K = gt["out_K"]
u, icov, x, counts, index = dense_front_end_select(None, out["xyz_weight_logits"], out["xyz_weights_scale"], None, out["msk_vis_logits"],
                                                   "quantile_in_mask", quantile=0.2, sample=1)
floatbits.decode_selected_rows(out["xyz_noc_bin"], gt["bit_cnt"], index, counts, x, noc_scale=gt["noc_scale"], model_transform=gt["model_transform"])
sel = dict(weights=icov, index=index, min_count=4)
start, inl, bad = gpu_solver.solve_device(K, x, u, counts, reprojectionError=2.0, reproj_divisor=gt["out_pix_scale"], refine=False, select=sel)
rows = torch.where(bad, torch.zeros_like(counts), counts)


def timed(fn, reps=100):
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


def lm1():
    return pnp_ceres.solve_device(K, x, u, None, start, rows, max_iter_count=20, weight_mask=inl.view(torch.uint8), return_iters=True)


st1, _, ret1, it1 = lm1()
fu, fw, fx, fc, _ = sel["result"]


def lm2():
    return pnp_ceres.solve_device(K, fx, fu, fw, st1, fc, weights_are_icov=True, nan_to_num=True, return_iters=True)


st2, _, ret2, it2 = lm2()
print(f"points per object {int(counts.min())}..{int(counts.max())}, RANSAC inliers {int(fc.min())}..{int(fc.max())}")
for name, it, ret, fn in (("inlier refinement (unit weights)", it1, ret1, lm1), ("weighted solve on the inliers", it2, ret2, lm2)):
    itf = it.float()
    print(f"{name:36s} LM iterations mean {itf.mean():.2f} max {int(it.max())} invalid {int(ret.sum())}  launch {timed(fn):6.2f} us (stream order, event-timed)")
