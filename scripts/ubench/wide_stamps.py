#!/usr/bin/env python3
"""Phase clock of the wide LM solve (lc_pnp_lm_wide_kernel<false,*,4>: four wavefronts per pose, up to four correspondences per thread in
registers): diagnostic build -DLC_STAMPS, thread 0 of every workgroup, shader cycles summed over the solve.  Never quote its run time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

LIB = build.variant_path("stamps")
if "--build" in sys.argv or not os.path.exists(LIB):
    build.build_variant("stamps", ["-DLC_STAMPS"])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["LC_AMD_LIB"] = LIB

import numpy as np  # noqa: E402
import torch  # noqa: E402

from lc_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
P = _lib.ptr
B, N = 64, int(os.environ.get("LC_WIDE_N", "1024"))
for used in ((200, 350, 700, 1024) if N == 1024 else tuple(int(v) for v in os.environ.get("LC_WIDE_USED", str(N)).split(","))):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1, outlier_frac=0.0, noise_px=0.7).items()}
    counts = torch.full((B,), used, dtype=torch.int32, device=dev)
    states = torch.empty_like(b["start"]); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
    stamps = torch.zeros(B, 16, device=dev, dtype=torch.int32)
    for rep in range(3):
        ws = torch.zeros(int(lib.lc_pnp_lm_workspace_bytes(B, N)) or 1, device=dev, dtype=torch.uint8) if os.environ.get("LC_WIDE_SPLIT") == "1" else None
        rc = lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(b["inv_std"]), None, P(counts), P(b["start"]), P(states), P(tr), P(ret),
                                P(stamps), B, N, 50, 1e-6, 0, 0, P(ws), 0 if ws is None else ws.numel(), None)
        assert rc == 0
        torch.cuda.synchronize()
    ps = stamps.cpu().numpy().view(np.uint64).reshape(B, 8).astype(np.int64)
    iters = ps[:, 6]
    tot = ps[:, :6].sum(1) + ps[:, 7]
    print(f"\n{used} of {N} correspondences: total cycles/workgroup median {np.median(tot):.0f}; LM iterations mean {iters.mean():.2f} (max {iters.max()})")
    for i, nm in enumerate(["prologue (loads, quat->aa)", "LM algebra without LDL^T", "make_rot (sincos)", "accumulate J^T J", "block sum (4 waves)", "copy / finite check"]):
        print(f"  {nm:28s} {np.median(ps[:, i]):8.0f}  {100 * np.median(ps[:, i]) / np.median(tot):5.1f} %   per evaluation {np.median(ps[:, i] / (iters + 1)):7.0f}")
    print(f"  {'LDL^T solve':28s} {np.median(ps[:, 7]):8.0f}  {100 * np.median(ps[:, 7]) / np.median(tot):5.1f} %   per iteration  {np.median(ps[:, 7] / np.maximum(iters, 1)):7.0f}")
