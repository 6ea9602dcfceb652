#!/usr/bin/env python3
"""Diagnostic build (-DLC_STAMPS): shader cycles per phase of one loss workgroup at a dense shape, tiled and one-workgroup form.
usage: loss_stamps.py [B N]   Never quote this build's run time (the stamps drain the memory queues)."""
import glob
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
so = "/tmp/liblc_amd_diag.so"
srcs = sorted(glob.glob(os.path.join(ROOT, "lc_amd", "csrc", "*.hip")))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLC_STAMPS", *srcs, "-o", so])
os.environ["LC_AMD_LIB"] = so
from lc_amd import synth  # noqa: E402
from lc_amd.cov_mixed import loss_cov_mixed_fused  # noqa: E402

dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1024)
b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=0).items()}
names = ["set-up + own loads", "pass1+reduce3", "pass2+reduce2", "pass3 accumulate", "reduce48 / hand-off", "gauss-jordan", "stepA h_j",
         "stepB sqrt", "loss scalar", "reverse 6x6", "pass4 (backward)"]
for tiled in (False, True):
    for rep in range(3):
        out = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], want_aux=True, tiled=tiled)
        torch.cuda.synchronize()
    st = out[4].cpu().numpy().view(np.uint64).reshape(B, 20)[:, :12].astype(np.int64)
    d = np.diff(st, axis=1)
    tot = st[:, 11] - st[:, 0]
    print(f"B={B} N={N} {'tiled' if tiled else 'one-workgroup'}: total cycles/workgroup median {np.median(tot):.0f} (min {tot.min()}, max {tot.max()})")
    for i in range(11):
        print(f"  {names[i]:22s} {np.median(d[:, i]):8.0f}  {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")
