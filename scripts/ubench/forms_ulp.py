#!/usr/bin/env python3
"""Do two FORMS of the LC loss give the same fp32 outputs for every input?  The dense pose unit slices N = 1849 as 4 x 8 tiles, the
stand-alone launch as 8 x 4: same sums in the same order, two inlined copies of the walk body.  Counts differing output elements over
thousands of random cotangents (under hipcc's default -ffp-contract=fast: ~4 in 10^8, all 1 ulp; under =on, the library's build: none).
usage: forms_ulp.py [trials]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lc_amd import synth
from lc_amd.cov_mixed import loss_cov_mixed_fused
from lc_amd.fused import PoseUnit
dev = torch.device("cuda:0")
res = {}
for B, N in ((32, 1849), (32, 1024), (64, 1024), (3, 2048), (5, 700)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N).items()}
    unit = PoseUnit(B, N, dev)
    bad = {"loss": 0, "d_pts2d": 0, "d_inv_std": 0, "d_pts3d": 0}
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for seed in range(trials):
        go = (torch.rand(B, generator=torch.Generator().manual_seed(seed)) + 0.5).to(dev)
        loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
        unit(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], b["bbox_3d"], b["start"], grad_out=go)
        for k, a, c in (("loss", unit.loss, loss), ("d_pts2d", unit.d_pts2d, du), ("d_inv_std", unit.d_inv_std, ds), ("d_pts3d", unit.d_pts3d, dx)):
            if not torch.equal(a, c):
                bad[k] += int((a != c).sum())
    res[(B, N)] = bad
    print(B, N, "differing elements over", trials, "random cotangents:", bad, flush=True)
