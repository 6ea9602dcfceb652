#!/usr/bin/env python3
"""Sweep the keypoint-head memory-policy bits (TUNING build only: lc_debug_head_variant) inside the alternating fwd;bwd step.
bit0 nt stores (bwd gradient), bit1 nt loads (bwd), bit2 reverse map order (bwd), bit3 nt loads (fwd)."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.lib_path())
    if not hasattr(raw, "lc_debug_head_variant"):
        raise SystemExit("this sweep needs the tuning hook (commit dd139a8: lc_debug_head_variant); the shipped library has the "
                         "policy fixed (lc_head.hip: kHeadPolicyF32 / kHeadPolicy16) -- results: profiles/r02/head_policy.txt")
    P = _lib.ptr
    for dtype in (torch.float32, torch.bfloat16):
        M, H, W = 256 * 64, 64, 64
        logits = torch.randn(M, H, W, device=dev).to(dtype)
        mean, std, stats = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev), torch.empty(M, 4, device=dev)
        g_mean, g_std = torch.randn(M, 2, device=dev), torch.randn(M, 2, device=dev)
        g_in = torch.empty_like(logits)
        code = _lib.MAP_DTYPES[dtype]
        st = _lib.stream_ptr(dev)
        nbytes = 3 * logits.numel() * logits.element_size()
        for variant in range(16):
            raw.lc_debug_head_variant(variant)

            def fwd():
                assert lib.lc_softargmax2d_fwd(P(logits), code, M, H, W, 0, P(mean), P(std), P(stats), st) == 0

            def bwd():
                assert lib.lc_softargmax2d_bwd(P(logits), code, P(mean), P(std), P(stats), P(g_mean), P(g_std), M, H, W, 0, P(g_in), st) == 0
            for _ in range(5):
                fwd(); bwd()
            torch.cuda.synchronize()
            reps = 40
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
            t0 = time.perf_counter()
            for e in evs:
                e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
            tf, tb = med([e[0].elapsed_time(e[1]) for e in evs]), med([e[1].elapsed_time(e[2]) for e in evs])
            print(json.dumps({"dtype": str(dtype), "variant": variant, "fwd_us": tf * 1e3, "bwd_us": tb * 1e3, "step_us": wall * 1e6,
                              "step_TBps": nbytes / wall / 1e12}), flush=True)
    raw.lc_debug_head_variant(0)


if __name__ == "__main__":
    main()
