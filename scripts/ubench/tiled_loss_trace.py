#!/usr/bin/env python3
"""Kernel durations of the LC-loss forms at dense shapes, from rocprofv3 (event timing through the Python wrapper is host-bound
at these sizes).  Two roles:
  tiled_loss_trace.py run            the workload: 100 launches per form and shape (run it under rocprofv3 --kernel-trace)
  tiled_loss_trace.py report DIR     averages per (kernel, grid size) from the *_kernel_trace.csv under DIR"""
import csv
import glob
import os
import sys
from collections import defaultdict

SHAPES = [(32, 1024), (32, 1849), (64, 1024), (128, 1024), (256, 1024), (8, 1024), (1, 4096), (16, 4096), (64, 4096), (32, 512), (64, 2048)]

if sys.argv[1] == "run":
    import torch

    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    from lc_amd import synth, cov_mixed as cm

    dev = torch.device("cuda:0")
    for B, N in SHAPES:
        b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1).items()}
        args = (b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])
        for tiled in (False, True):
            for _ in range(100):
                cm.loss_cov_mixed_fused(*args, tiled=tiled)
            torch.cuda.synchronize()
else:
    ev = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lc_cov_loss" in r["Kernel_Name"]:
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "tiled" in r["Kernel_Name"],
                           int(r.get("Grid_Size_X") or r["Grid_Size"]) // int(r.get("Workgroup_Size_X") or r["Workgroup_Size"])))
    ev.sort()
    assert len(ev) == 200 * len(SHAPES), len(ev)
    for i, (B, N) in enumerate(SHAPES):  # launch order: per shape 100 x one-workgroup form, then 100 x tiled request
        one, til = ev[200 * i:200 * i + 100], ev[200 * i + 100:200 * i + 200]
        avg = lambda v: sum(x[1] for x in v) / len(v) / 1e3  # noqa: E731
        print(f"B={B:4d} N={N:5d}  one-workgroup {avg(one):7.1f} us ({one[0][3]} workgroups)   "
              f"{'tiled' if til[0][2] else 'tiled not offered ->'} {avg(til):7.1f} us ({til[0][3]} workgroups)")
