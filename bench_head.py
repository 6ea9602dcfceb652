#!/usr/bin/env python3
"""Keypoint-head benchmark: spatial softmax + soft-argmax forward+backward over logits (256,64,64,64) fp32
(268 MB) resident in HBM.  HBM-bound: algorithmic traffic = read 1 MiB/sample forward + read 1 MiB + write 1 MiB backward.
Prints one JSON line; bench.py imports measure_head() to attach the same numbers to its own line."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def in_training_head_times():
    """The head kernels' rocprofv3 averages INSIDE a training step (sparse example, B=256, 64 maps of 64x64 in bf16, freshly written by the
    backbone), read from the newest committed `profiles/<round>/g1/sparse_kernel_stats.csv` -- never a literal.  None if no such file."""
    import csv
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "g1", "sparse_kernel_stats.csv")), reverse=True):
        fwd = bwd = None
        for r in csv.DictReader(open(path)):
            if "lc_head_fwd" in r["Name"]:
                fwd = float(r["AverageNs"]) / 1e3
            elif "lc_head_bwd" in r["Name"]:
                bwd = float(r["AverageNs"]) / 1e3
        if fwd and bwd:
            by = 256 * 64 * 64 * 64 * 2  # bf16 maps
            return {"fwd_us": round(fwd, 2), "bwd_us": round(bwd, 2), "fwd_frac_of_hbm": round(by / fwd / 1e3 / HBM_PEAK_GBS, 3),
                    "bwd_frac_of_hbm": round(2 * by / bwd / 1e3 / HBM_PEAK_GBS, 3),
                    "source": os.path.relpath(path, ROOT) + " (rocprofv3 --kernel-trace --stats of examples/train_sparse_ddp.py, B=256, bf16 maps)"}
    return None


def measure_head(dev, B=256, S=64, H=64, W=64, steps=20, warmup=3, dtype="f32"):
    from lc_amd import _lib

    lib = _lib.load()
    P = _lib.ptr
    g = torch.Generator(device="cpu").manual_seed(0)
    M = B * S
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    code, esz = _lib.MAP_DTYPES[tdt], torch.empty((), dtype=tdt).element_size()
    logits = torch.randn(M, H, W, generator=g).to(tdt).to(dev)  # synthetic logits of the repo's shape (values do not change traffic)
    mean = torch.empty(M, 2, device=dev)
    std = torch.empty(M, 2, device=dev)
    stats = torch.empty(M, 4, device=dev)
    g_mean = torch.randn(M, 2, generator=g).to(dev)
    g_std = torch.randn(M, 2, generator=g).to(dev)
    g_in = torch.empty_like(logits)
    st = _lib.stream_ptr(dev)

    def fwd():
        assert lib.lc_softargmax2d_fwd(P(logits), code, M, H, W, 0, P(mean), P(std), P(stats), st) == 0

    def bwd():
        assert lib.lc_softargmax2d_bwd(P(logits), code, P(mean), P(std), P(stats), P(g_mean), P(g_std), M, H, W, 0, P(g_in), st) == 0

    def step():
        fwd()
        bwd()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    wins = []
    for _ in range(5):  # median of five windows of `steps` steps (occasional ~60 ms stalls on the shared pool)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        wins.append(time.perf_counter() - t0)
    el = sorted(wins)[len(wins) // 2]

    # in-pattern kernel times: events around each kernel INSIDE the alternating fwd;bwd loop (a forward that follows the
    # backward's 268 MB of dirty lines is not the forward that re-reads a cache-warm buffer) -- medians over the steps
    reps = max(steps, 20)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for e in evs:
        e[0].record()
        fwd()
        e[1].record()
        bwd()
        e[2].record()
    torch.cuda.synchronize(dev)
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    t_f = med([e[0].elapsed_time(e[1]) for e in evs])
    t_b = med([e[1].elapsed_time(e[2]) for e in evs])

    def ev(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    bb_f, bb_b = ev(fwd), ev(bwd)  # the same kernel launched back to back over the same buffer (Infinity-Cache assisted: NOT the step)
    map_bytes = H * W * esz
    by_f, by_b = M * map_bytes, 2 * M * map_bytes
    step_gbs = (by_f + by_b) / (el / steps) / 1e9
    gbs = lambda by, ms: by / (ms * 1e-3) / 1e9  # noqa: E731
    return {
        "metric": "keypoint-head samples/sec (spatial softmax + soft-argmax fwd+bwd)",
        "value": B * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3,
        "dtype": dtype,
        "config": {"workload": f"logits ({B},{S},{H},{W}) {dtype}, synthetic"},
        "roofline": {"bound": "hbm", "kernel": "lc_head_bwd_kernel", "achieved": gbs(by_b, t_b), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": gbs(by_b, t_b) / HBM_PEAK_GBS, "traffic": None,
                     "timing": "events around each kernel inside the alternating fwd;bwd loop (median)",
                     "cache_note": "same-buffer, Infinity-Cache-assisted: the loop re-reads ONE 268 MB logits buffer with a 256 MiB Infinity "
                                   "Cache behind it (non-temporal gradient stores leave part of it resident), which is how the step gets above "
                                   "the 6.29 TB/s copy ceiling; inside a training step the maps are freshly written by the backbone: "
                                   "`in_training_rocprof_bf16` holds the kernels' averages there (from the newest committed g1 profile) -- "
                                   "quote those with any use of this figure",
                     "in_training_rocprof_bf16": in_training_head_times(),
                     "fwd": {"kernel": "lc_head_fwd_wave64_kernel" if (H, W) == (64, 64) else "lc_head_fwd_rows_kernel",
                             "achieved": gbs(by_f, t_f), "frac": gbs(by_f, t_f) / HBM_PEAK_GBS, "ms": t_f},
                     "bwd_ms": t_b,
                     "step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS, "note": "wall-clock fwd+bwd step, all algorithmic bytes"},
                     "back_to_back_same_buffer": {"fwd_ms": bb_f, "fwd_frac": gbs(by_f, bb_f) / HBM_PEAK_GBS, "bwd_ms": bb_b,
                                                  "bwd_frac": gbs(by_b, bb_b) / HBM_PEAK_GBS,
                                                  "note": "cache-assisted best case, not what a training step sees"},
                     "algorithmic_bytes_per_sample": 3 * S * map_bytes},
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"], help="element type of the logits and their gradient")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    print(json.dumps(measure_head(torch.device("cuda:0"), B=a.batch, steps=a.steps, warmup=a.warmup, dtype=a.dtype)))
