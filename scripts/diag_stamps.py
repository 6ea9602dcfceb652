#!/usr/bin/env python3
"""Diagnostic build (-DLC_STAMPS): where does one loss workgroup spend its cycles?  Builds a SEPARATE library in /tmp, runs
the B=256 N=64 loss launch, prints the median s_memtime delta per phase (shader cycles).  Never quote this build's run time."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/liblc_amd_diag.so"
import glob
srcs = sorted(glob.glob(os.path.join(ROOT, "lc_amd", "csrc", "*.hip")))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLC_STAMPS", *srcs, "-o", so])
os.environ["LC_AMD_LIB"] = so
from lc_amd import synth  # noqa: E402
from lc_amd.cov_mixed import loss_cov_mixed_fused  # noqa: E402

dev = torch.device("cuda:0")
B, N = 256, 64
b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=0).items()}
names = ["load pose/K", "load pts", "pass1+reduce3", "pass2+reduce2", "pass3 accumulate", "reduce48+expand", "gauss-jordan", "stepA h_j",
         "stepB sqrt", "loss scalar", "reverse 6x6", "pass4"]
for rep in range(3):
    out = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], want_aux=True)
    torch.cuda.synchronize()
st = out[4].cpu().numpy().view(np.uint64).reshape(B, 20)[:, :12].astype(np.int64)
d = np.diff(st, axis=1)
tot = st[:, 11] - st[:, 0]
print(f"total cycles/workgroup: median {np.median(tot):.0f}  (min {tot.min()}, max {tot.max()})")
for i in range(11):
    print(f"  {names[i + 1]:22s} {np.median(d[:, i]):8.0f}  {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")

# ---- PnP LM kernel: phase sums over the whole solve ----
lib = __import__("lc_amd._lib", fromlist=["x"]).load()
from lc_amd import _lib  # noqa: E402
states = torch.empty_like(b["start"]); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
stamps = torch.zeros(B, 16, device=dev, dtype=torch.int32)  # 8 x uint64 per pose
P = _lib.ptr
for rep in range(3):
    rc = lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(b["inv_std"]), None, None, P(b["start"]), P(states), P(tr), P(ret), P(stamps), B, N, 50, 1e-6, 0, 0, None, 0, None)
    torch.cuda.synchronize()
ps = stamps.cpu().numpy().view(np.uint64).reshape(B, 8).astype(np.int64)
iters = ps[:, 6]
tot = ps[:, :6].sum(1) + ps[:, 7]
print(f"\nPnP: total cycles/wave median {np.median(tot):.0f}; LM iterations mean {iters.mean():.2f} (min {iters.min()}, max {iters.max()})")
for i, nm in enumerate(["prologue (loads, quat->aa)", "LM algebra without LDL^T", "make_rot (sincos)", "accumulate J^T J", "reduce-scatter 32", "LDS broadcast"]):
    print(f"  {nm:28s} {np.median(ps[:, i]):8.0f}  {100 * np.median(ps[:, i]) / np.median(tot):5.1f} %   per evaluation {np.median(ps[:, i] / (iters + 1)):7.0f}")
print(f"  {'LDL^T solve (of the algebra)':28s} {np.median(ps[:, 7]):8.0f}  {100 * np.median(ps[:, 7]) / np.median(tot):5.1f} %   per iteration  {np.median(ps[:, 7] / np.maximum(iters, 1)):7.0f}")
