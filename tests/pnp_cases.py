"""Seeded weighted-PnP problem sets shared by the golden generators and the parity tests (inputs only; everything comes
from lc_amd.synth, so a generator on another machine rebuilds bit-identical inputs and the fixtures can be checked for that)."""
import numpy as np
import torch

from lc_amd import synth


def _perturbed_start(b, rot, trans, g):
    B = b["pose"].shape[0]
    rv = torch.randn(B, 3, generator=g) * rot
    ang = rv.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    dq = torch.cat(((ang / 2).cos(), rv / ang * (ang / 2).sin()), -1)
    qs = synth._quat_mul(b["pose"][:, :4], dq)
    return torch.cat((qs, b["pose"][:, 4:] * (1 + trans * torch.randn(B, 3, generator=g))), -1).float().contiguous()


def pnp_case(name):
    """-> dict(K (B,3,3), pts3d (B,N,3), pts2d (B,N,2), sqrtL (B,N,2,2), start (B,7), counts (B,) int32, max_iter, ftol), float32."""
    g = torch.Generator().manual_seed(1234)
    kw = dict(max_iter=50, ftol=1e-6)
    if name == "metric_B256_N64":          # BASELINE.json configs[1]
        b = synth.make_batch(256, 64, seed=0)
        start = b["start"]
    elif name == "hard_B512_N12":          # long trajectories, rejected steps, NO_CONVERGENCE exits
        b = synth.make_batch(512, 12, seed=31, outlier_frac=0.2, noise_px=2.0)
        start = _perturbed_start(b, 0.3, 0.1, g)
    elif name == "minimal_B256_N4":        # near-minimal problems (P3P-like ambiguity)
        b = synth.make_batch(256, 4, seed=41, outlier_frac=0.0, noise_px=0.5)
        start = _perturbed_start(b, 0.08, 0.03, g)
    elif name == "dense_B16_N1024":        # glmo / gycbv dense shape
        b = synth.make_batch(16, 1024, seed=51, outlier_frac=0.1, noise_px=1.0)
        start = b["start"]
    elif name == "ragged_full_B64_N48":    # ragged counts incl. < 3 points, genuine 2x2 lower factors
        b = synth.make_batch(64, 48, seed=61)
        start = b["start"]
    elif name == "identity_B64_N24":       # small-angle branches of the three rotation conversions
        b = synth.make_batch(64, 24, seed=3, noise_px=0.5, outlier_frac=0.0, rotate_K=False)
        q = torch.cat((torch.ones(64, 1), torch.randn(64, 3, generator=g) * 1e-9), -1)
        q[:8, 1:] = 0
        q = q / q.norm(dim=-1, keepdim=True)
        pose = torch.cat((q, b["pose"][:, 4:]), -1)
        R = synth._quat_to_R(q.double())
        xf = (b["pts3d"].double() @ R.mT + pose[:, None, 4:].double()) @ b["K"].double().mT
        b["pts2d"] = (xf[..., :2] / xf[..., 2:3] + 0.3 * torch.randn(64, 24, 2, generator=g).double()).float()
        start = pose.clone()
        start[:, 4:] *= 1.01
    elif name == "maxiter1_B32_N64":       # NO_CONVERGENCE contract: every job invalid, states untouched
        b = synth.make_batch(32, 64, seed=71)
        start = b["start"]
        kw["max_iter"] = 1
    else:
        raise KeyError(name)
    B, N = b["pts3d"].shape[:2]
    L = torch.diag_embed(b["inv_std"])
    counts = torch.full((B,), N, dtype=torch.int32)
    if name == "ragged_full_B64_N48":
        L[..., 1, 0] = (torch.rand(B, N, generator=g) - 0.5) * 0.6
        counts = torch.randint(3, N + 1, (B,), generator=g).int()
        counts[5], counts[17], counts[40] = 2, 0, 1
    out = dict(K=b["K"], pts3d=b["pts3d"], pts2d=b["pts2d"], sqrtL=L, start=start.float(), counts=counts)
    out = {k: np.ascontiguousarray(v.numpy()) for k, v in out.items()}
    out.update(kw)
    return out


PNP_CASES = ("metric_B256_N64", "hard_B512_N12", "minimal_B256_N4", "dense_B16_N1024", "ragged_full_B64_N48", "identity_B64_N24",
             "maxiter1_B32_N64")


def pose_err(a, b):
    """max|dq| after sign alignment, ||dt||/||t|| per job (SURVEY 8c tolerance definition)."""
    qa = a[:, :4] / np.linalg.norm(a[:, :4], axis=1, keepdims=True)
    qb = b[:, :4] / np.linalg.norm(b[:, :4], axis=1, keepdims=True)
    sgn = np.sign((qa * qb).sum(1, keepdims=True))
    sgn[sgn == 0] = 1
    return np.abs(qa - sgn * qb).max(1), np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1) / np.linalg.norm(b[:, 4:], axis=1)
