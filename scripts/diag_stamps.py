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
srcs = [os.path.join(ROOT, "lc_amd", "csrc", f) for f in ("lc_capi.hip", "lc_fused.hip", "lc_head.hip", "lc_loss.hip", "lc_pnp.hip")]
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
