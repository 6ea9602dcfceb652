#!/usr/bin/env python3
"""Soak test of the tiled LC-loss hand-off under UNEVEN load (MI355X_MICROARCH.md: idle chips and uniform load hide stale reads):
thousands of tiled launches over a few shapes on two streams while a third stream keeps every CU streaming through HBM (the keypoint
head's forward / backward over 268 MB) and a fourth runs the one-wave pose-unit launch; every output of every launch is compared bit
for bit with the one-workgroup form's.  Prints the number of launches checked and of mismatches (must be 0).
usage: tiled_soak.py [seconds]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import _lib, synth, cov_mixed as cm  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
P = _lib.ptr
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
shapes = [(32, 1024), (32, 1849), (16, 4096), (64, 2048), (7, 700), (1, 4096)]
cases = []
for i, (B, N) in enumerate(shapes):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=40 + i, outlier_frac=0.1).items()}
    args = (b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])
    ref = cm.loss_cov_mixed_fused(*args, tiled=False)
    cases.append((args, ref))
# background load: the keypoint head over 268 MB (HBM-bound, every CU busy) and the pose unit (512 one-wave workgroups)
M = 256 * 64
logits = torch.randn(M, 64, 64, device=dev)
mean, std, stats = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev), torch.empty(M, 4, device=dev)
gm, gs, gin = torch.randn(M, 2, device=dev), torch.randn(M, 2, device=dev), torch.empty_like(logits)
pu = {k: v.to(dev) for k, v in synth.make_batch(256, 64, seed=3).items()}
pu_out = [torch.empty(256, device=dev), torch.empty_like(pu["pts2d"]), torch.empty_like(pu["pts2d"]), torch.empty_like(pu["pts3d"]),
          torch.empty_like(pu["start"]), torch.empty(256, device=dev), torch.empty(256, device=dev, dtype=torch.int32)]
go = torch.full((256,), 1 / 256, device=dev)
s_head, s_unit, s_a, s_b = (torch.cuda.Stream(dev) for _ in range(4))
torch.cuda.synchronize()
checked = bad = rounds = 0
t0 = time.time()
while time.time() - t0 < budget:
    rounds += 1
    with torch.cuda.stream(s_head):
        for _ in range(3):
            st = _lib.stream_ptr(dev)
            lib.lc_softargmax2d_fwd(P(logits), 0, M, 64, 64, 0, P(mean), P(std), P(stats), st)
            lib.lc_softargmax2d_bwd(P(logits), 0, P(mean), P(std), P(stats), P(gm), P(gs), M, 64, 64, 0, P(gin), st)
    with torch.cuda.stream(s_unit):
        for _ in range(10):
            lib.lc_pose_unit2_f32(P(pu["K"]), P(pu["pose"]), P(pu["pts3d"]), P(pu["pts2d"]), P(pu["inv_std"]), None, P(pu["bbox_3d"]), P(go), 256, 64, 32.0, 3.0, 4.0, P(pu_out[0]), P(pu_out[1]), P(pu_out[2]), P(pu_out[3]), P(pu["inv_std"]), P(pu["start"]), P(pu_out[4]), P(pu_out[5]), P(pu_out[6]), None, 50, 1e-6, None, 0, _lib.stream_ptr(dev))
    outs = []
    for j in range(12):
        args, ref = cases[(rounds + j) % len(cases)]
        with torch.cuda.stream(s_a if j % 2 else s_b):
            outs.append((cm.loss_cov_mixed_fused(*args), ref))
    torch.cuda.synchronize()
    for got, ref in outs:
        checked += 1
        if not all(torch.equal(a, c) for a, c in zip(got[:4], ref[:4])):
            bad += 1
print(f"tiled soak: {rounds} rounds, {checked} tiled launches checked against the one-workgroup form under head + pose-unit load, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
