#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference ABI `pnp_ceres_f32_omp` (host pointer arrays in, GPU body): profiles/r03/NOTES.md section 5."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lc_amd import _lib, synth  # noqa: E402

lib = ctypes.CDLL(_lib.lib_path())
fp = ctypes.POINTER(ctypes.c_float)
for B in (256, 4096):
    b = synth.make_batch(B, 64, seed=0)
    L = np.ascontiguousarray(torch.diag_embed(b["inv_std"]).numpy())
    arrs = {k: np.ascontiguousarray(b[k].numpy()) for k in ("K", "pts2d", "pts3d", "start")}
    PA = fp * B
    ptr = lambda a: PA(*[a[i].ctypes.data_as(fp) for i in range(B)])
    counts = np.full(B, 64, np.int32)
    tr, ret = np.zeros(B, np.float32), np.zeros(B, np.int32)

    st = arrs["start"].copy()
    tabs = (ptr(st), ptr(arrs["K"]), ptr(arrs["pts2d"]), ptr(arrs["pts3d"]), ptr(L))  # pointer tables built once: the C side is timed

    def call():
        st[:] = arrs["start"]
        lib.pnp_ceres_f32_omp(*tabs, counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                              ctypes.c_int(50), ctypes.c_float(1e-6), ctypes.c_int(0), tr.ctypes.data_as(fp),
                              ret.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.c_int(B), ctypes.c_int(4))

    for _ in range(3):
        call()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        call()
    dt = (time.perf_counter() - t0) / n
    print(f"pnp_ceres_f32_omp B={B} N=64: {dt * 1e6:.0f} us per call (gather to pinned staging, H2D, kernel, D2H, scatter) -> {B / dt / 1e6:.2f} M poses/s")
