#!/usr/bin/env python3
"""Summarise scripts/profile_g1.sh: per-kernel table of a real-width training step, share of GPU time per kernel family,
MFMA utilisation of the conv/GEMM kernels (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles =
GRBM_GUI_ACTIVE / 8 because rocprofv3 sums that counter over the 8 XCDs -- MI355X_MICROARCH.md), eager vs GraphedLoss wall clock."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def family(name):
    n = name.lower()
    if name.startswith("lc_") or "lc::" in name or "lc_" in name.split("(")[0]:
        return "lc_* (this repo's HIP kernels)"
    if "nccl" in n or "rccl" in n:
        return "RCCL"
    if any(k in n for k in ("igemm", "conv", "cijk", "gemm", "xdlops", "winograd", "miopen", "implicit", "naive_conv", "sp3asm", "gridwise", "wrw", "fwd_", "bwd_")) and "batchnorm" not in n and "elementwise" not in n:
        return "conv / GEMM (MIOpen, hipBLASLt, Tensile)"
    if "batchnorm" in n or "batch_norm" in n or "bn" in n.split("_")[:2]:
        return "batch norm"
    if "at::native" in name or "elementwise" in n or "vectorized" in n or "reduce" in n or "copy" in n or "fill" in n or "cat" in n:
        return "PyTorch elementwise / reduce / copy"
    return "other"


def short(name):
    m = re.search(r"lc_\w+?_kernel", name)
    if m:
        return m.group(0)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return name[-70:]


def first(pattern):
    f = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return f[0] if f else None


print("# g1: real-width training step around the hot path (one MI355X, bf16 autocast backbone, synthetic data)\n")
def hbm_table(rows, B=32, S=128, C=21, e=2, sample=3):
    """The lc_* kernels of the zlmo-shaped step that stream maps: algorithmic bytes (what the operation must move once) / time / 8 TB/s."""
    HW, N = S * S, (-(-S // sample)) ** 2
    algo = {  # kernel -> (bytes, what)
        "lc_xyz_bin_loss_fwd_kernel": (B * C * HW * (e + 1) + B * HW * e, "code logits + target bits + visibility logits, once"),
        "lc_xyz_bin_loss_bwd_kernel": (B * C * HW * (e + 1 + e) + B * HW * e, "the same + the logits' gradient"),
        "lc_xyz_bin_loss_bwd_plane_kernel": (B * C * HW * (e + 1 + e) + B * HW * e, "the same + the logits' gradient"),
        "lc_bits_decode_gt_fwd_kernel": (B * N * (C * (e + 1) + 1 + 12), "sampled pixels only: C logits + C raw bits + mask in, 3 floats out (whole sampled rows "
                                         "at sector granularity: %.1f MB)" % (B * C * (-(-S // sample)) * S * (e + 1) / 1e6)),
        "lc_bits_decode_gt_bwd_kernel": (B * C * HW * e + B * N * (C * (e + 1) + 1 + 12), "the logits' gradient map (zero off the sampled pixels) + the forward's reads + the cotangent"),
        "lc_bits_decode_gt_bwd_tile_kernel": (B * C * HW * e + B * N * (C * (e + 1) + 1 + 12), "the logits' gradient map (zero off the sampled pixels) + the forward's reads + the cotangent"),
        "lc_bits_decode_gt_fwd_wide_kernel": (B * N * (C * (e + 1) + 1 + 12), "sampled pixels only: C logits + C raw bits + mask in, 3 floats out (latency-shaped: one memory round trip)"),
        "lc_dense_frontend_fwd_kernel": (B * 2 * HW * e + B * N * 16, "weight logits (joint softmax over 2HW) in, pts2d + inv_std out"),
        "lc_dense_frontend_bwd_kernel": (B * 2 * HW * e * 2 + B * N * 8, "weight logits in, their gradient out, the cotangent of inv_std"),
        "lc_dense_aux_fwd_kernel": (B * HW * (e + 4), "visibility logits + mask"),
        "lc_dense_aux_bwd_kernel": (B * HW * (e + 4 + e), "the same + the gradient"),
        "lc_sqnorm_kernel": (B * 2 * HW * e, "the weight logits' gradient"),
        "lc_clip_apply_kernel": (B * 2 * HW * e * 2, "the gradient in and out"),
    }
    print(f"HBM view of the lc_* kernels that stream maps (B={B}, {S}x{S} maps, {C} planes, {e}-byte elements, N={N}); algorithmic bytes = what the operation must move once:\n")
    print("| kernel | avg us | algorithmic MB | GB/s | of 8 TB/s | bytes counted |")
    print("|---|---|---|---|---|---|")
    merged = {}
    for r in rows:  # template instances of one kernel (one per sub-sampling phase's tile geometry): one row, weighted
        k = short(r["Name"])
        c, t = merged.get(k, (0, 0.0))
        merged[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    for k, (calls, total) in merged.items():
        if k in algo:
            by, what = algo[k]
            us = total / calls / 1e3
            print(f"| {k} | {us:.1f} | {by / 1e6:.2f} | {by / us / 1e3:.0f} | {100 * by / us / 1e3 / 8000:.1f} % | {what} |")
    print()


for tag, title in (("zlmo", "binary-code head at zlmo's own shape (configs[4], configs/zlmo.yaml): output-stride-8 dilated ResNet-34-width encoder + atrous pyramid + skip "
                            "decoder (examples/os8_trunk.py), fp16 autocast + GradScaler, B=32, 128x128 maps, 7+7+7 code planes, dense_sample 3 => N=1849"),
                   ("zlmo2", "the same step SHARDED: two ranks of B=32 on this one GPU over gloo (rank 0 profiled) -- which launches a sharded job adds, not how long they "
                             "take (the ranks' kernels share the GPU and gloo stages every collective through the host)"),
                   ("dense", "dense head (glmo shape, configs[2]): ResNet-34-width encoder + decoder, B=32, 64x64 maps, N=1024 correspondences"),
                   ("sparse", "sparse head: same backbone, B=256, 64 keypoint maps of 64x64 (the metric's B=256, N=64 inside a training step)")):
    print(f"## {title}\n")
    for mode in ("eager", "graphs"):
        f = os.path.join(out, f"{tag}_{mode}.log")
        if os.path.exists(f):
            m = [ln.strip() for ln in open(f) if ln.startswith("median step")]
            d = [ln.strip() for ln in open(f) if ln.startswith("quartiles of the step time")]
            print(f"* wall clock, Loss_fn {mode}: {m[-1] if m else 'no result (see log)'}{' [' + d[-1] + ']' if d else ''}")
    print()
    if tag == "zlmo2":
        for ln in open(os.path.join(out, "zlmo_2rank_r0.log")) if os.path.exists(os.path.join(out, "zlmo_2rank_r0.log")) else []:
            if ln.startswith(("median step", "comm summary")):
                print("* " + ln.strip())
        print()
    f = first(f"{tag}_trace/**/*kernel_stats.csv")
    if not f:
        print("(no kernel stats)\n")
        continue
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = defaultdict(float)
    for r in rows:
        fam[family(r["Name"])] += float(r["TotalDurationNs"])
    print("| kernel family | GPU time share | ms over the run |")
    print("|---|---|---|")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {100 * v / tot:.2f} % | {v / 1e6:.2f} |")
    print(f"\nAll kernels: {tot / 1e6:.1f} ms over 14 steps (incl. 2 warm-up steps).\n")
    print("| kernel (top 14 by time + every lc_*) | family | calls | avg us | % |")
    print("|---|---|---|---|---|")
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for i, r in enumerate(rows):
        if i < 14 or family(r["Name"]).startswith("lc_"):
            print(f"| {short(r['Name'])} | {family(r['Name']).split(' ')[0]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    print()
    if tag == "zlmo":
        hbm_table(rows)
    f = first(f"{tag}_pmc/**/*counter_collection.csv")
    if f:
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[r["Kernel_Name"]] += 1
        mf = [(k, d) for k, d in acc.items() if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and d.get("GRBM_GUI_ACTIVE", 0) > 0]
        mf.sort(key=lambda kd: -kd[1]["GRBM_GUI_ACTIVE"])
        tot_busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for _, d in mf)
        tot_cyc = sum(d["GRBM_GUI_ACTIVE"] for _, d in mf)
        all_cyc = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in acc.values())
        print("MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), kernels that issue MFMA:\n")
        print("| kernel | dispatches | MFMA utilisation | share of all GPU cycles |")
        print("|---|---|---|---|")
        for k, d in mf[:12]:
            print(f"| {short(k)} | {cnt[k]} | {100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * d['GRBM_GUI_ACTIVE']):.1f} % | {100 * d['GRBM_GUI_ACTIVE'] / all_cyc:.1f} % |")
        if tot_cyc:
            print(f"\nAll MFMA kernels together: {100 * tot_busy / (128 * tot_cyc):.1f} % MFMA utilisation over {100 * tot_cyc / all_cyc:.1f} % of the GPU cycles; "
                  f"whole step: {100 * tot_busy / (128 * all_cyc):.1f} %.\n")
