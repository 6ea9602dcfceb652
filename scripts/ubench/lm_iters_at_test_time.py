#!/usr/bin/env python3
"""LM iteration counts of the two solves of the test-time pipeline (64 objects, 64x64 maps) and their event-timed launch durations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.dense import dense_front_end_select  # noqa: E402
from lc_amd.pnp import gpu_solver, pnp_ceres  # noqa: E402
from lc_amd.synth import dense_inputs  # noqa: E402

dev = torch.device("cuda:0")
gt, out = dense_inputs(B=64, H=64, W=64, seed=3)
out["xyz_weight_logits"] = out["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
out["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
gt = {k: v.to(dev) for k, v in gt.items()}
out = {k: v.to(dev).contiguous() for k, v in out.items()}
K = gt["out_K"]
u, icov, x, counts, index = dense_front_end_select(out["xyz_noc"], out["xyz_weight_logits"], out["xyz_weights_scale"], gt["noc_scale"],
                                                   out["msk_vis_logits"], "quantile_in_mask", quantile=0.5, sample=2)
sel = dict(weights=icov, index=index, min_count=4)
start, inl, bad = gpu_solver.solve_device(K, x, u, counts, reprojectionError=3.0, refine=False, select=sel)
rows = torch.where(bad, torch.zeros_like(counts), counts)


def timed(fn, reps=200):
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
U2, W2, X2, C2 = torch.cat((u, fu)), torch.cat((icov, fw)), torch.cat((x, fx)), torch.cat((counts, fc))


def lm2():
    return pnp_ceres.solve_device(K, X2, U2, W2, st1, C2, weights_are_icov=True, nan_to_num=True, shared_poses=64, return_iters=True)


st2, _, ret2, it2 = lm2()
print(f"points per object {int(counts.min())}..{int(counts.max())}, RANSAC inliers {int(fc.min())}..{int(fc.max())}")
for name, it, fn in (("inlier refinement (64 poses)", it1, lm1), ("weighted + weighted-filtered (128 poses)", it2, lm2)):
    itf = it.float()
    print(f"{name:42s} LM iterations mean {itf.mean():.2f} max {int(it.max())}   launch {timed(fn):6.2f} us (stream order, event-timed)")
