#!/usr/bin/env python3
"""Soak of the deterministic last-block reductions (lc_dense_aux_fwd, lc_xyz_bin_loss_fwd, lc_sqnorm: per-block partials written through
the caches, a relaxed arrival counter, the last block adds them in block order): the same inputs on two streams, thousands of launches,
a streaming kernel on a third stream as background load -- every result must equal the first bit for bit.  usage: reduction_soak.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import dense_aux  # noqa: E402
from lc_amd.grad import NormClipper  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
g = torch.Generator().manual_seed(0)
shapes = [(32, 64, 64, 17), (7, 37, 45, 21), (64, 128, 128, 21)]
data = []
for B, H, W, C in shapes:
    data.append(dict(xyz=torch.randn(B, 3, H, W, generator=g).to(dev), tgt=torch.randn(B, 3, H, W, generator=g).to(dev),
                     mn=(torch.rand(B, H, W, generator=g) > 0.4).to(dev), sl=torch.randn(B, 1, H, W, generator=g).to(dev),
                     mv=(torch.rand(B, H, W, generator=g) > 0.5).float().to(dev), wl=torch.randn(B, 2, H, W, generator=g).to(dev),
                     lg=(torch.randn(B, C, H, W, generator=g) * 3).to(dev), bits=(torch.rand(B, C, H, W, generator=g) > 0.5).to(dev), C=C))


def run(d):
    aux = torch.stack(dense_aux.dense_aux_losses(d["xyz"], d["mn"], d["tgt"], d["sl"], d["mv"], d["wl"], "bce"))
    hist = torch.full((d["C"],), 0.5, device=dev)
    bl = dense_aux.xyz_bin_loss(d["lg"], d["bits"], d["sl"], hist, 0.05)
    clip = NormClipper().to(dev)
    clipped = clip.clip(d["wl"])
    return torch.cat((aux, bl[None], hist, clip.max_norm.reshape(1), clipped.flatten()[:64]))


want = [run(d).clone() for d in data]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
load = torch.cuda.Stream(dev)
big = torch.randn(64 * 1024 * 1024 // 4, device=dev)
t0, rounds, checked, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    with torch.cuda.stream(load):
        for _ in range(8):
            big.mul_(1.0000001)
    outs = []
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            for i, d in enumerate(data):
                outs.append((i, run(d)))
    torch.cuda.synchronize()
    for i, o in outs:
        checked += 1
        bad += 0 if torch.equal(o, want[i]) else 1
    rounds += 1
print(f"reduction soak: {rounds} rounds, {checked} result sets (3 reductions each) checked against the first under streaming load on another stream, "
      f"{bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
