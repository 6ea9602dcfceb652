#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (not part of pytest: minutes, not seconds): the HIP loss and PnP kernels against
the oracle over many random shapes, noise levels, outlier rates, start perturbations, ragged counts and masks.  Prints
one line per configuration and a summary; `profiles/<round>/fuzz_parity.txt` keeps the last run.

    python tests/fuzz_parity.py [--cases 60] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
from lc_amd import synth  # noqa: E402
from lc_amd.cov_mixed import _launch_loss  # noqa: E402
from lc_amd.pnp import pnp_ceres  # noqa: E402
from oracle import lc_loss_oracle, pnp_oracle  # noqa: E402

DEV = torch.device("cuda:0")


def pose_err(a, b):
    qa = a[:, :4] / np.linalg.norm(a[:, :4], axis=1, keepdims=True)
    qb = b[:, :4] / np.linalg.norm(b[:, :4], axis=1, keepdims=True)
    sgn = np.sign((qa * qb).sum(1, keepdims=True))
    return np.abs(qa - sgn * qb).max(1), np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1) / np.linalg.norm(b[:, 4:], axis=1)


def perturbed_start(b, rot, trans, g):
    B = b["pose"].shape[0]
    rv = torch.randn(B, 3, generator=g) * rot
    ang = rv.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    dq = torch.cat(((ang / 2).cos(), rv / ang * (ang / 2).sin()), -1)
    q = b["pose"][:, :4]
    qs = torch.stack((q[:, 0] * dq[:, 0] - (q[:, 1:] * dq[:, 1:]).sum(-1),
                      q[:, 0] * dq[:, 1] + q[:, 1] * dq[:, 0] + q[:, 2] * dq[:, 3] - q[:, 3] * dq[:, 2],
                      q[:, 0] * dq[:, 2] - q[:, 1] * dq[:, 3] + q[:, 2] * dq[:, 0] + q[:, 3] * dq[:, 1],
                      q[:, 0] * dq[:, 3] + q[:, 1] * dq[:, 2] - q[:, 2] * dq[:, 1] + q[:, 3] * dq[:, 0]), -1)
    return torch.cat((qs, b["pose"][:, 4:] * (1 + trans * torch.randn(B, 3, generator=g))), -1).float().contiguous()


def fuzz_pnp(case, rng, g):
    N = int(rng.choice([3, 4, 6, 12, 33, 64, 65, 100, 300, 1024]))
    B = 2048 if N <= 64 else 256
    noise, outl = float(rng.choice([0.0, 0.5, 2.0])), float(rng.choice([0.0, 0.05, 0.3]))
    rot, trans = [(0.02, 0.01), (0.08, 0.03), (0.3, 0.1)][int(rng.integers(3))]
    full = bool(rng.integers(2))
    ragged = bool(rng.integers(2))
    b = synth.make_batch(B, N, seed=1000 + case, outlier_frac=outl, noise_px=noise)
    start = perturbed_start(b, rot, trans, g)
    if full:  # a genuine lower-triangular factor
        L = torch.diag_embed(b["inv_std"])
        L[..., 1, 0] = (torch.rand(B, N, generator=g) - 0.5) * 0.6
    else:
        L = b["inv_std"]
    counts = torch.randint(2, N + 1, (B,), generator=g).int() if ragged else None
    st, tr, ret, iters = pnp_ceres.solve_device(b["K"].to(DEV), b["pts3d"].to(DEV), b["pts2d"].to(DEV), L.to(DEV), start.to(DEV),
                                                None if counts is None else counts.to(DEV), return_iters=True)
    Lfull = L if full else torch.diag_embed(L)
    so, tro, reto = pnp_oracle.solve_batched(start.numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(), Lfull.numpy(),
                                             counts=None if counts is None else counts.numpy(), num_threads=32)
    ret = ret.cpu().numpy()
    same = ret == reto
    both = (ret == 0) & (reto == 0)
    dq, dt = pose_err(st.cpu().numpy()[both], so[both]) if both.any() else (np.zeros(1), np.zeros(1))
    within = ((dq <= 1e-4) & (dt <= 1e-4)).mean()
    untouched = np.array_equal(st.cpu().numpy()[ret == 1], start.numpy()[ret == 1])
    print(f"pnp  case {case:3d} B={B:4d} N={N:3d} noise={noise} outl={outl} start=({rot},{trans}) full={int(full)} ragged={int(ragged)} | "
          f"iters max {int(iters.max())} invalid {int((ret == 1).sum())} | flags equal {same.mean():.4f} within 1e-4 {within:.4f} "
          f"p99 dq {np.quantile(dq, 0.99):.1e} dt {np.quantile(dt, 0.99):.1e} | invalid keep start: {untouched}", flush=True)
    return same.mean(), within, untouched, B


def fuzz_loss(case, rng, g):
    N = int(rng.choice([3, 5, 16, 64, 65, 130, 256, 257, 700, 1024, 1849]))  # > 256: the tiled form (small B) and the loop form
    B = int(rng.integers(1, 48))
    noise, outl = float(rng.choice([0.2, 1.0, 4.0])), float(rng.choice([0.0, 0.05, 0.3]))
    cov2d = bool(rng.integers(4) == 0)
    b = synth.make_batch(B, N, seed=5000 + case, outlier_frac=outl, noise_px=noise)
    valid = None
    if rng.integers(2):
        valid = (torch.rand(B, N, generator=g) > 0.25).float()
        valid[:, :3] = 1
    go = torch.rand(B, generator=g) + 0.5
    d = {k: v.to(DEV) for k, v in b.items()}
    loss, gu, gs, gx, _ = _launch_loss(d["K"], d["pose"], d["pts3d"], d["pts2d"], d["inv_std"], None if valid is None else valid.to(DEV),
                                       d["bbox_3d"], go.to(DEV), 32.0, 3.0, 4.0, True, True, cov_2d=cov2d)
    b64 = {k: v.double() for k, v in b.items()}
    rl, ru, rs, rx = lc_loss_oracle.loss_and_grads(b64["K"], b64["pose"], b64["pts3d"], b64["pts2d"], b64["inv_std"],
                                                   None if valid is None else valid.double(), b64["bbox_3d"], grad_out=go.double(),
                                                   want_pts3d=True, cov_2d=cov2d)
    rel = lambda a, r: ((a.cpu().double() - r).abs().max() / r.abs().max().clamp_min(1e-300)).item()  # noqa: E731
    el = ((loss.cpu().double() - rl).abs() / rl.abs().clamp_min(1)).max().item()
    eu, es, ex = rel(gu, ru), rel(gs, rs), rel(gx, rx)
    print(f"loss case {case:3d} B={B:3d} N={N:3d} noise={noise} outl={outl} valid={int(valid is not None)} cov2d={int(cov2d)} | "
          f"loss {el:.1e} d_pts2d {eu:.1e} d_inv_std {es:.1e} d_pts3d {ex:.1e}", flush=True)
    return max(el, 0.0), max(eu, es, ex)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    pn, ls = [], []
    for c in range(a.cases):
        pn.append(fuzz_pnp(c, rng, g))
        ls.append(fuzz_loss(c, rng, g))
    jobs = sum(p[3] for p in pn)
    print(f"SUMMARY pnp: {len(pn)} configurations, {jobs} jobs; validity flags equal in {np.average([p[0] for p in pn], weights=[p[3] for p in pn]):.5f} "
          f"of jobs (worst configuration {min(p[0] for p in pn):.4f}); accepted poses within 1e-4: mean {np.mean([p[1] for p in pn]):.5f}, "
          f"worst {min(p[1] for p in pn):.4f}; invalid jobs keep their start in all configurations: {all(p[2] for p in pn)}")
    print(f"SUMMARY loss: {len(ls)} configurations; worst loss error {max(l[0] for l in ls):.2e} (rel, floor 1), worst gradient error "
          f"{max(l[1] for l in ls):.2e} (rel to max |grad|)")


if __name__ == "__main__":
    main()
