#!/usr/bin/env python3
"""(Lives under tests/ like fuzz_parity.py: it uses the oracle as its checker, which only tests may import.)
Soak of the claim "the RANSAC's integer outputs are exact for EVERY pose": random batches (poses, row length, ragged counts, noise, gross outlier
share, iterations, per-pose thresholds), both launch forms, every checked pose against oracle/p3p_ransac_oracle.py: ransac_f32 on the kernel's
own float32 hypotheses -- winner, inlier count, inlier mask, validity, per-hypothesis counts and the bits of the per-hypothesis inlier error.
    python tests/soak_ransac_exact.py [batches=60] [seed=1]      (on the MI355X; ~3 min)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lc_amd import synth  # noqa: E402
from lc_amd.pnp import gpu_solver  # noqa: E402
from oracle import p3p_ransac_oracle as O  # noqa: E402

batches, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
poses = hyps = bad_pose = bad_hyp = ties = 0
shapes = []
t0 = time.time()
for it in range(batches):
    N = int([16, 16, 40, 64, 300, 1024, 3300, 5000, 16384][int(torch.randint(0, 9, (1,), generator=g))])
    B = int(torch.randint(2, 9 if N > 2000 else 40, (1,), generator=g))
    noise = float(torch.rand(1, generator=g)) * 1.2 * (0 if it % 7 == 3 else 1)  # every seventh batch noise-free: the ERROR decides the winner
    outl = float(torch.rand(1, generator=g)) * 0.4 * (0 if it % 7 == 3 else 1)
    iters = int([64, 150, 192, 300][int(torch.randint(0, 4, (1,), generator=g))])
    b = synth.make_batch(B, N, seed=1000 * seed + it, outlier_frac=outl, noise_px=noise)
    counts = torch.randint(max(3, N // 3), N + 1, (B,), generator=g).to(torch.int32)
    thr = torch.rand(B, generator=g) * 2.5 + 0.5
    args = (b["K"].to(dev), b["pts3d"].to(dev), b["pts2d"].to(dev), counts.to(dev))
    ws = []
    kw = dict(reprojectionError=thr.to(dev), iterations=iters, seed=it, refine=False, return_hypothesis=True)
    split = gpu_solver.solve_device(*args, split=True, workspace_out=ws, **kw)
    single = gpu_solver.solve_device(*args, split=False, **kw)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(split, single)), ("launch forms differ", it, B, N)
    st, inl, bad, hyp, n_in = (t.cpu().numpy() for t in split)
    hyp64, hyp32, pc, pe = (v.cpu().numpy() for v in gpu_solver.workspace_views(ws[0], B, N, iters))
    rows = range(B) if N <= 1024 else range(min(B, 3))
    for r_ in rows:
        n = int(counts[r_])
        r = O.ransac_f32(b["K"][r_].numpy(), b["pts3d"][r_].numpy(), b["pts2d"][r_].numpy(), n, float(thr[r_]), hyp32[r_])
        poses += 1
        ok = (int(bad[r_]) == r["invalid"] and int(hyp[r_]) == r["best_hyp"] and int(n_in[r_]) == r["n_inliers"]
              and np.array_equal(inl[r_].astype(bool), r["inlier_mask"]))
        bad_pose += not ok
        if r["per_hyp_count"] is not None:
            C = (n + 63) // 64
            cnt = pc[r_, :C].astype(np.int64).sum(0)
            err = np.zeros(pe.shape[-1], np.float32)
            for c in range(C):
                err = (err + pe[r_, c]).astype(np.float32)
            hyps += len(cnt)
            bad_hyp += int((cnt != r["per_hyp_count"]).sum() + (err.view(np.uint32) != r["per_hyp_err"].view(np.uint32)).sum())
            top = r["per_hyp_count"].max()
            ties += int((r["per_hyp_count"] == top).sum() > 1)
    shapes.append((B, N, iters))
print(f"{batches} batches (seed {seed}), shapes N in {sorted(set(s[1] for s in shapes))}: {poses} poses and {hyps} hypotheses checked in {time.time() - t0:.0f} s")
print(f"poses whose winner / inlier count / inlier mask / validity differ from the float32 oracle: {bad_pose}")
print(f"hypotheses whose inlier count or inlier-error BITS differ: {bad_hyp}")
print(f"poses where several hypotheses share the top count (the error sum decided): {ties}")
sys.exit(1 if (bad_pose or bad_hyp) else 0)
