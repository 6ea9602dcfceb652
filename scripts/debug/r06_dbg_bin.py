import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from lc_amd import dense_aux
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(4)
B, C, H, W = 6, 21, 32, 32
logits = (torch.randn(B, C, H, W, generator=g) * 2).to(torch.float16)
bits = (logits > 0) ^ (torch.rand(B, C, H, W, generator=g) < 0.25)
vis = torch.randn(B, 1, H, W, generator=g).to(torch.float16)
logits, bits, vis = logits.to(dev), bits.to(dev), vis.to(dev)
for step in range(3):
    hist = torch.full((C,), 0.5, device=dev)
    x = (logits * (1 + 0.3 * step)).clone().requires_grad_(True)
    loss = dense_aux.xyz_bin_loss(x, bits, vis, hist, 0.05)
    (loss * 4096).backward()
    gfull = x.grad.float()
    for lo, hi in ((0, 3), (3, 6)):
        hist2 = torch.full((C,), 0.5, device=dev)
        xr = (logits[lo:hi] * (1 + 0.3 * step)).clone().requires_grad_(True)
        print("x equal", torch.equal(xr.detach(), x.detach()[lo:hi]))
        l2 = dense_aux.xyz_bin_loss(xr, bits[lo:hi], vis[lo:hi], hist2, 0.05)
        (l2 * 4096).backward()
        gr = xr.grad.float() / 2
        # weights differ here (own histogram) -> compare ratio per channel
        ratio = (gr / gfull[lo:hi])
        for c in range(C):
            r = ratio[:, c][torch.isfinite(ratio[:, c]) & (gfull[lo:hi][:, c].abs() > 1e-4)]
            if r.numel():
                rr = r / r.median()
                bad = (rr - 1).abs() > 2e-3
                if bad.any():
                    idx = bad.nonzero()[:3]
                    print("step", step, "shard", lo, "ch", c, "bad", int(bad.sum()), "of", r.numel(), rr[bad][:3].tolist())
