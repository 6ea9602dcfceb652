#!/usr/bin/env python3
"""How often does `rets` (the integer output of the weighted-PnP solve) differ from the DENSE_QR oracle when only the
linear solver of the LM step changes?  CPU only; same problem distribution as tests/fuzz_parity.py.

    python scripts/pnp_numerics/flip_rates.py [--cases 60] [--seed 0]
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lc_amd import synth  # noqa: E402
from fuzz_parity import perturbed_start  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(ROOT, "oracle", "_build", "libpnp_variants.so")
KINDS = {0: "qr (oracle)", 1: "qr, reversed sums", 2: "normal eq + LDLt", 3: "normal eq + 1 refinement", 4: "normal eq in double-double"}


def lib():
    src = os.path.join(HERE, "variants.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "pnp_lm_oracle.c"))):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-o", SO, src, "-lm"])
    L = ctypes.CDLL(SO)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.pnp_oracle_batched_f32.argtypes = [fp, fp, fp, fp, fp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_float, fp, ip, ctypes.c_int, ctypes.c_int]
    return L


def solve(L, kind, start, K, u, X, sqrtL, counts, threads):
    L.variants_set_kind(kind)
    st = np.ascontiguousarray(start, np.float32).copy()
    B, N = X.shape[:2]
    tr, ret = np.zeros(B, np.float32), np.zeros(B, np.int32)
    f = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
    L.pnp_oracle_batched_f32(f(st), f(K), f(u), f(X), f(sqrtL), counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), N, 50, 1e-6,
                             f(tr), ret.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), B, threads)
    return st, tr, ret


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    L = lib()
    rng = np.random.default_rng(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    flips = {k: 0 for k in KINDS}
    far = {k: 0 for k in KINDS}
    jobs = 0
    for case in range(a.cases):
        N = int(rng.choice([3, 4, 6, 12, 33, 64, 65, 100, 300]))
        B = 2048 if N <= 64 else 256
        noise, outl = float(rng.choice([0.0, 0.5, 2.0])), float(rng.choice([0.0, 0.05, 0.3]))
        rot, trans = [(0.02, 0.01), (0.08, 0.03), (0.3, 0.1)][int(rng.integers(3))]
        full, ragged = bool(rng.integers(2)), bool(rng.integers(2))
        b = synth.make_batch(B, N, seed=1000 + case, outlier_frac=outl, noise_px=noise)
        start = perturbed_start(b, rot, trans, g)
        Lf = torch.diag_embed(b["inv_std"])
        if full:
            Lf[..., 1, 0] = (torch.rand(B, N, generator=g) - 0.5) * 0.6
        counts = (torch.randint(2, N + 1, (B,), generator=g).int() if ragged else torch.full((B,), N, dtype=torch.int32)).numpy()
        args = [np.ascontiguousarray(t.numpy(), np.float32) for t in (start, b["K"], b["pts2d"], b["pts3d"], Lf)]
        ref = solve(L, 0, *args, counts, a.threads)
        line = f"case {case:3d} B={B} N={N:3d} noise={noise} outl={outl} start=({rot},{trans}) full={int(full)} ragged={int(ragged)} invalid {int(ref[2].sum()):4d} |"
        for k in KINDS:
            if k == 0:
                continue
            st, tr, ret = solve(L, k, *args, counts, a.threads)
            nf = int((ret != ref[2]).sum())
            both = (ret == 0) & (ref[2] == 0)
            d = np.abs(st[both] - ref[0][both])
            nfar = int(((d[:, :4].max(1) > 1e-4) | (np.linalg.norm(d[:, 4:], axis=1) > 1e-4 * np.linalg.norm(ref[0][both][:, 4:], axis=1))).sum()) if both.any() else 0
            flips[k] += nf
            far[k] += nfar
            line += f" k{k}: {nf}/{nfar}"
        jobs += B
        print(line, flush=True)
    print(f"SUMMARY {jobs} jobs; per variant: flag flips / accepted poses further than 1e-4 from the DENSE_QR oracle")
    for k in KINDS:
        if k:
            print(f"  {KINDS[k]:28s}: {flips[k]:5d} ({flips[k] / jobs:.5%})  / {far[k]:5d} ({far[k] / jobs:.5%})")


if __name__ == "__main__":
    main()
