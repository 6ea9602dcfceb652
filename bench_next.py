#!/usr/bin/env python3
"""Measurement of the SURVEY.md 8f "next" rows (f1 dense front end, f2 RANSAC initialiser, f3 ZebraPose codes, f4 pose-error
metrics) and of the wide (N > 64) PnP on one MI355X.  Not the headline (bench.py is): one JSON line per kernel with the
event-timed launch duration and the algorithmic bytes it moves, priced against the 8 TB/s HBM peak where the kernel is a
streaming one; the compute-shaped ones (RANSAC, ADI nearest neighbour, LM) carry a note instead of a meaningful fraction."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def ev(fn, dev, reps, warm=3):
    """Median over `reps` launches of the event-bracketed duration (a median, because a one-off host stall of tens of ms --
    allocator or runtime housekeeping -- otherwise lands inside a back-to-back window; rocprofv3's per-kernel average in
    profiles/ is the authoritative figure)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize(dev)
    d = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return d[len(d) // 2] * 1e3  # us


def line(name, us, nbytes, units, unit_name, note=None, **cfg):
    gbs = nbytes / (us * 1e-6) / 1e9
    rec = {"kernel": name, "us": round(us, 2), "units_per_s": units / (us * 1e-6), "unit": unit_name + "/s",
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": nbytes}, "config": cfg}
    if note:
        rec["roofline"]["note"] = note
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    dev = torch.device("cuda:0")
    from lc_amd import _lib, dense, floatbits, metrics, synth
    from lc_amd.pnp import gpu_solver, pnp_ceres

    lib = _lib.load()
    P, st = _lib.ptr, _lib.stream_ptr(dev)
    g = torch.Generator(device="cpu").manual_seed(0)

    # ---- f1: dense front end, (B,3,128,128) maps, stride-2 sub-sampling -> N = 4096 points per sample ----
    B, H, W, sample = 64, 128, 128, 2
    xyz = torch.randn(B, 3, H, W, generator=g).to(dev)
    wl = torch.randn(B, 2, H, W, generator=g).to(dev)
    ws = torch.rand(B, generator=g).to(dev) + 0.5
    ns = torch.rand(B, 3, generator=g).to(dev) + 0.5
    N = (H // sample) * (W // sample)
    out = dense._launch_fwd(xyz, wl, ws, ns, 0, 0, sample)
    us = ev(lambda: dense._launch_fwd(xyz, wl, ws, ns, 0, 0, sample), dev, a.reps)
    # the joint softmax normaliser reads every weight logit (2 maps), the strided subset of xyz is gathered, N*(2+2+3) floats out
    by = B * (2 * H * W * 4 + 3 * N * 4 + 7 * N * 4)
    line("lc_dense_frontend_fwd_kernel", us, by, B, "samples", B=B, H=H, W=W, sample=sample)
    lse = out[3]
    if True:
        gi = torch.randn(B, N, 2, generator=g).to(dev)
        gp = torch.randn(B, N, 3, generator=g).to(dev)
        need = (True, True, True)
        us = ev(lambda: dense._launch_bwd(wl, ws, ns, lse, gi, gp, (B, H, W), 0, 0, sample, need), dev, a.reps)
        by = B * (2 * H * W * 4 + 5 * N * 4 + 5 * H * W * 4)  # re-read logits + grads in, dense (zero-filled) d_xyz + d_wlogits out
        line("lc_dense_frontend_bwd_kernel", us, by, B, "samples", B=B, H=H, W=W, sample=sample)

    # ---- f1 (test-time half): point selection + compaction of the N = 4096 dense correspondences ----
    pts2d_d, inv_std_d, pts3d_d, _ = out
    seg = (torch.rand(B, N, generator=g) > 0.4).to(torch.uint8).to(dev)
    for mode in ("mask", "quantile", "quantile_in_mask"):
        us = ev(lambda: dense.dense_select(pts2d_d, inv_std_d, pts3d_d, mode, mask=seg, quantile=0.7), dev, a.reps)
        line("lc_dense_select_kernel (%s)" % mode, us, B * N * (28 + 1) + B * int(N * 0.5) * 32, B, "samples",
             note="one workgroup per sample; quantile modes radix-select the two order statistics from keys staged in LDS; the time here is "
                  "the Python call (output allocation included), the kernel alone is in next_kernel_stats.csv", B=B, N=N, mode=mode)

    # ---- f1, test time at 64x64 maps (1024 sampled pixels per object): front end + selection in ONE launch ----
    Hs = Ws = 64
    xyz_s, wl_s = torch.randn(B, 3, Hs, Ws, generator=g).to(dev), torch.randn(B, 2, Hs, Ws, generator=g).to(dev)
    vl_s = torch.randn(B, 1, Hs, Ws, generator=g).to(dev)
    for mode in ("mask", "quantile_in_mask"):
        us = ev(lambda: dense.dense_front_end_select(xyz_s, wl_s, ws, ns, vl_s, mode, quantile=0.5, sample=2), dev, a.reps)
        line("lc_dense_frontend_select_kernel (%s)" % mode, us, B * (6 * Hs * Ws * 4 + 512 * 32), B, "samples",
             note="front end + point selection of test.py:85-113 in one launch, one workgroup per object; replaces lc_dense_frontend_fwd_kernel + "
                  "lc_dense_select_kernel when an object has at most 1024 sampled pixels", B=B, H=Hs, W=Ws, sample=2, mode=mode)

    # ---- f3: ZebraPose codes: 3x7-bit logits over 128x128 ----
    C, bits = 21, 7
    lg = torch.randn(B, C, H, W, generator=g).to(dev)
    gb = (torch.rand(B, C, H, W, generator=g) > 0.5).to(torch.uint8).to(dev)
    gm = (torch.rand(B, H, W, generator=g) > 0.3).to(torch.uint8).to(dev)
    us = ev(lambda: floatbits.nn_logits2noc(lg, bits), dev, a.reps)
    line("lc_bits_decode_kernel", us, B * H * W * (C * 4 + 12), B, "samples", B=B, C=C, H=H, W=W)
    b3 = floatbits._bits3(bits, C)
    us = ev(lambda: floatbits._launch_decode_gt(lg, gb, gm, b3, 0, 0, 1), dev, a.reps)
    line("lc_bits_decode_gt_fwd_kernel", us, B * H * W * (C * 4 + C + 1 + 12), B, "samples", B=B, C=C, H=H, W=W)
    gn = torch.randn(B, H * W, 3, generator=g).to(dev)
    us = ev(lambda: floatbits._launch_decode_gt_bwd(lg, gb, gm, gn, b3, 0, 0, 1, True), dev, a.reps)
    line("lc_bits_decode_gt_bwd_kernel", us, B * H * W * (2 * C * 4 + C + 1 + 12), B, "samples", B=B, C=C, H=H, W=W)

    # the same decodes with the callers' coordinate map applied in the kernel (noc_scale, model transform; the inference form writes planes)
    scl = (torch.rand(B, 3, generator=g) * 100 + 20).to(dev)
    Tm = torch.eye(4).repeat(B, 1, 1).to(dev)
    us = ev(lambda: floatbits.nn_logits2xyz_planes(lg, bits, scl, Tm), dev, a.reps)
    line("lc_bits_decode_kernel (planes, scale + model transform)", us, B * H * W * (C * 4 + 12), B, "samples", B=B, C=C, H=H, W=W,
         note="replaces decode + broadcast multiply + subtract + batched GEMM + permute-copy")

    # ---- f1 / f3 on the maps a mixed-precision backbone emits (BASELINE.json configs[2] bf16, configs[4] fp16): read in their own type, no
    # up-cast copy in front (round 3: a 132 MB cast pass each way around a 124 MB kernel at this shape); bytes priced at 2 per map element ----
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        lg16 = lg.to(dt)
        us = ev(lambda: floatbits.nn_logits2noc(lg16, bits), dev, a.reps)
        line(f"lc_bits_decode_kernel [{tag} logits]", us, B * H * W * (C * 2 + 12), B, "samples", B=B, C=C, H=H, W=W, map_dtype=tag)
        us = ev(lambda: floatbits.nn_logits2xyz_planes(lg16, bits, scl, Tm), dev, a.reps)
        line(f"lc_bits_decode_kernel (planes, scale + model transform) [{tag} logits]", us, B * H * W * (C * 2 + 12), B, "samples", B=B, C=C, H=H, W=W, map_dtype=tag)
        us = ev(lambda: floatbits._launch_decode_gt(lg16, gb, gm, b3, 0, 0, 1), dev, a.reps)
        line(f"lc_bits_decode_gt_fwd_kernel [{tag} logits]", us, B * H * W * (C * 2 + C + 1 + 12), B, "samples", B=B, C=C, H=H, W=W, map_dtype=tag)
        us = ev(lambda: floatbits._launch_decode_gt_bwd(lg16, gb, gm, gn, b3, 0, 0, 1, True), dev, a.reps)
        line(f"lc_bits_decode_gt_bwd_kernel [{tag} logits, {tag} gradient]", us, B * H * W * (2 * C * 2 + C + 1 + 12), B, "samples", B=B, C=C, H=H, W=W, map_dtype=tag)
        xyz16, wl16 = xyz.to(dt), wl.to(dt)
        o16 = dense._launch_fwd(xyz16, wl16, ws, ns, 0, 0, sample)
        us = ev(lambda: dense._launch_fwd(xyz16, wl16, ws, ns, 0, 0, sample), dev, a.reps)
        line(f"lc_dense_frontend_fwd_kernel [{tag} maps]", us, B * (2 * H * W * 2 + 3 * N * 2 + 7 * N * 4), B, "samples", B=B, H=H, W=W, sample=sample, map_dtype=tag)
        us = ev(lambda: dense._launch_bwd(wl16, ws, ns, o16[3], gi, gp, (B, H, W), 0, 0, sample, (True, True, True)), dev, a.reps)
        line(f"lc_dense_frontend_bwd_kernel [{tag} maps, {tag} gradients]", us, B * (2 * H * W * 2 + 5 * N * 4 + 5 * H * W * 2), B, "samples", B=B, H=H, W=W,
             sample=sample, map_dtype=tag)
        vl16 = torch.randn(B, 1, H, W, generator=g).to(dev).to(dt)
        us = ev(lambda: dense.dense_front_end_select(xyz16, wl16, ws, ns, vl16, "quantile_in_mask", quantile=0.2, sample=1), dev, a.reps)
        line(f"lc_dense_frontend_select_kernel (quantile_in_mask, 16384 candidates per object) [{tag} maps]", us, B * (6 * H * W * 2 + int(0.4 * H * W) * 32), B, "samples",
             note="zlmo's test-time shape (configs/zlmo.yaml:30-37): front end + selection of 128x128 candidates per object in one launch; latency-shaped "
                  "(one workgroup per object, radix select of 16384 keys in LDS), not bandwidth-shaped", B=B, H=H, W=W, sample=1, map_dtype=tag)

    # ---- a19, dense heads: the auxiliary losses of Loss_fn (one launch each way) and the code loss (one pass over the logits) ----
    from lc_amd import dense_aux
    xyz_a = torch.randn(B, 3, H, W, generator=g).to(dev)
    tgt_a = torch.randn(B, 3, H, W, generator=g).to(dev)
    mn_a = (torch.rand(B, H, W, generator=g) > 0.4).to(dev)
    sl_a = torch.randn(B, 1, H, W, generator=g).to(dev)
    mv_a = (torch.rand(B, H, W, generator=g) > 0.5).float().to(dev)
    wl_a = torch.randn(B, 2, H, W, generator=g).to(dev)
    us = ev(lambda: dense_aux.dense_aux_losses(xyz_a, mn_a, tgt_a, sl_a, mv_a, wl_a, "bce"), dev, a.reps)
    line("lc_dense_aux_fwd_kernel", us, B * H * W * (3 * 4 + 3 * 4 + 1 + 4 + 4 + 2 * 4), B, "samples", B=B, H=H, W=W,
         note="loss_noc + loss_seg + loss_weight_seg (losses.py:281-316) from one pass; ~12 torch launches in the reference")
    xg, sg, wg = xyz_a.clone().requires_grad_(True), sl_a.clone().requires_grad_(True), wl_a.clone().requires_grad_(True)
    l3 = dense_aux.dense_aux_losses(xg, mn_a, tgt_a, sg, mv_a, wg, "bce")
    tot = l3[0] + l3[1] + l3[2]
    us = ev(lambda: torch.autograd.grad(tot, (xg, sg, wg), retain_graph=True), dev, a.reps)
    line("lc_dense_aux_bwd_kernel", us, B * H * W * (2 * (3 * 4 + 4 + 2 * 4) + 3 * 4 + 1 + 4), B, "samples", B=B, H=H, W=W,
         note="timed through autograd (three scalar adds' backward included)")
    hist = torch.full((C,), 0.5, device=dev)
    gbb = gb.view(torch.bool)
    us = ev(lambda: dense_aux.xyz_bin_loss(lg, gbb, sl_a, hist, 0.05), dev, a.reps)
    line("lc_xyz_bin_loss_fwd_kernel", us, B * H * W * (C * 4 + C + 4), B, "samples", B=B, C=C, H=H, W=W,
         note="Loss_xyz_bin (losses.py:196-216) in one pass over the code logits; ~12 passes in the reference; VALU-bound (one exp + one log per logit) "
              "with a serial tail (per-channel partials -> histogram EMA -> softmax weights)")
    lgg = lg.clone().requires_grad_(True)
    lb = dense_aux.xyz_bin_loss(lgg, gbb, sl_a, hist, 0.05)
    us = ev(lambda: torch.autograd.grad(lb, lgg, retain_graph=True), dev, a.reps)
    line("lc_xyz_bin_loss_bwd_kernel", us, B * H * W * (2 * C * 4 + C + 4), B, "samples", B=B, C=C, H=H, W=W)

    # ---- f4: pose errors over a test set: 1024 poses x 2048 model points (ADI is an all-pairs nearest neighbour) ----
    Bp, M = 1024, 2048
    bt = synth.make_batch(Bp, 8, seed=1)
    from lc_amd.transforms import quaternion_rep_to_RT
    Rg, tg = quaternion_rep_to_RT(bt["pose"].to(dev))
    Re, te = quaternion_rep_to_RT(bt["start"].to(dev))
    pts = (torch.rand(M, 3, generator=g) - 0.5).to(dev) * 0.2
    us = ev(lambda: metrics.compute_pose_errors(Re, te, Rg, tg, pts), dev, max(3, a.reps // 5))
    line("lc_pose_errors_kernel (add+adi)", us, Bp * (24 * 4 + 16) + M * 12, Bp, "poses",
         note="VALU-bound, not HBM: %.2f T point pairs/s = %.0f %% of the packed-fp32 issue bound (3.5 VALU instructions per "
              "pair; 1024 SIMDs x 16 lanes/clk x 2.4 GHz / 3.5 = 11.2 T pairs/s)" % (Bp * M * M / (us * 1e-6) / 1e12,
                                                                                    Bp * M * M / (us * 1e-6) / 11.2e12 * 100), B=Bp, M=M)
    us = ev(lambda: metrics.compute_pose_errors(Re, te, Rg, tg, pts, want_adi=False), dev, a.reps)
    line("lc_pose_errors_kernel (add only)", us, Bp * (24 * 4 + 16) + M * 12, Bp, "poses", B=Bp, M=M)

    # ---- f2: RANSAC initialiser ----
    for (Br, Nr) in ((256, 64), (64, 1024), (256, 1024)):
        bt = synth.make_batch(Br, Nr, seed=2, outlier_frac=0.2)
        K, X, U = bt["K"].to(dev), bt["pts3d"].to(dev), bt["pts2d"].to(dev)
        for split, label in ((False, "lc_pnp_ransac_kernel (single launch)"), (True, "lc_ransac_{hypotheses,score,select}_kernel (split form)")):
            us = ev(lambda: gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, split=split), dev, a.reps)
            line(label, us, Br * (Nr * 21 + 36 + 40), Br, "poses",
                 note="compute-shaped: 192 P3P hypotheses x N reprojections per pose (%.1f G reprojections/s); gpu_solver picks the form from "
                      "the shape (split when N ceil(B/256) > 256)" % (Br * 192 * Nr / (us * 1e-6) / 1e9), B=Br, N=Nr, hypotheses=192)

    # ---- LC loss on the dense heads' shapes (training step of the dense configs): N = 1024 / 4096 points per sample ----
    from lc_amd import cov_mixed
    for (Bl, Nl) in ((32, 1024), (64, 1024), (64, 4096), (256, 1024)):
        bt = synth.make_batch(Bl, Nl, seed=4)
        K, pose, X, U, Wt, bb = (bt[k].to(dev) for k in ("K", "pose", "pts3d", "pts2d", "inv_std", "bbox_3d"))
        us = ev(lambda: cov_mixed._launch_loss(K, pose, X, U, Wt, None, bb, None, 32.0, 0.3, 8.0, True, True), dev, a.reps)
        line("lc_cov_loss_kernel (dense shape, fwd+bwd incl. d_pts3d)", us, Bl * (Nl * (28 + 28) + 36 + 28 + 96 + 4), Bl, "samples",
             note="%.1f M points/s" % (Bl * Nl / us), B=Bl, N=Nl)

    # ---- test-time driver end to end (test.py:67-136): network outputs -> poses, every stage on the device ----
    import time
    from lc_amd.config import AttrDict
    from lc_amd.inference import solve_pnp
    from lc_amd.synth import dense_inputs
    gt_d, out_d = dense_inputs(B=64, H=64, W=64, seed=3)
    out_d["xyz_weight_logits"] = out_d["xyz_weight_logits"] + 3 * gt_d["msk_vis"][:, None]
    out_d["msk_vis_logits"] = (gt_d["msk_vis"][:, None] * 2 - 1) * 4
    gt_d = {k: v.to(dev) for k, v in gt_d.items()}
    out_d = {k: v.to(dev).contiguous() for k, v in out_d.items()}  # a network's output is contiguous NCHW (the generator's xyz map is a strided view)
    cfg = AttrDict(dense_point_select="quantile_in_mask", quantile=0.5, dense_sample=2, solvers=["weighted", "weighted_filtered"])
    for _ in range(3):
        solve_pnp(cfg, out_d, gt_d)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        solve_pnp(cfg, out_d, gt_d)
    torch.cuda.synchronize(dev)
    us = (time.perf_counter() - t0) / a.reps * 1e6
    line("inference.solve_pnp_dense (front end + select, RANSAC + inlier re-selection, refinement + 2 weighted solves: 5 launches)", us,
         64 * 6 * 64 * 64 * 4, 64, "objects", note="wall clock per call incl. Python; no host synchronisation inside the pipeline",
         B=64, H=64, W=64, N=1024, select="quantile_in_mask")

    # ---- wide PnP (dense heads): N = 1024 and 1849 ----
    for Nw in (1024, 1849):
        bt = synth.make_batch(256, Nw, seed=3)
        K, X, U, Wt, S0 = (bt[k].to(dev) for k in ("K", "pts3d", "pts2d", "inv_std", "start"))
        us = ev(lambda: pnp_ceres.solve_device(K, X, U, Wt, S0), dev, a.reps)
        line("lc_pnp_lm_wide_kernel", us, 256 * (Nw * 28 + 36 + 28 + 36), 256, "poses",
             note="latency/VALU-bound LM iterations (see DESIGN.md 4, lc_pnp_lm_kernel)", B=256, N=Nw)


if __name__ == "__main__":
    main()
