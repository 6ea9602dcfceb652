#!/usr/bin/env python3
"""Fixtures for the GPU PnP initialiser: inputs + the outputs of oracle/p3p_ransac_oracle.py (an independent float64 P3P over the
kernel's own hypothesis stream).  OpenCV (cv2.solvePnPRansac, lib/pnp/cv2_solver.py:69-88) is absent from the image and its
RNG is not reproducible, so these are ORACLE vectors, not reference vectors; they freeze the integer outputs -- best hypothesis
index, inlier count, inlier index set -- for the poses the oracle marks as decided (the winner's count lead exceeds the points it and
its rivals have within 1e-3 of the threshold); the inlier mask is exact outside `mask_unsure`.

    python tests/golden/gen_golden_ransac.py [case ...]     (build container, CPU, ~1 min)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lc_amd import synth  # noqa: E402
from oracle import p3p_ransac_oracle as O  # noqa: E402

CASES = {  # name: (B, N, seed, noise_px, gross outlier fraction, reprojectionError, iterations, ransac seed)
    "clean_B16_N32": (16, 32, 5, 0.0, 0.0, 0.5, 150, 0),
    "outliers_B24_N64": (24, 64, 6, 0.5, 0.25, 2.0, 150, 0),
    "dense_B6_N400": (6, 400, 8, 1.0, 0.4, 3.0, 192, 7),
    "ragged_B8_N40": (8, 40, 7, 0.2, 0.1, 2.0, 150, 3),
    "wide_B6_N2300": (6, 2300, 9, 0.3, 0.3, 2.5, 192, 4),  # more points than one LDS tile of the single launch / 36 chunks of the split form
    # the first 2500 points of every row are gross outliers: the consensus lives in the tail of the row (a selection compacted in
    # raster order puts the object's lower rows there), so an implementation that samples or scores a prefix only finds nothing
    "tail_B4_N5000": (4, 5000, 11, 0.3, 0.2, 2.5, 192, 5),
}


def make_inputs(name):
    B, N, seed, noise, outl, thr, iters, rseed = CASES[name]
    b = synth.make_batch(B, N, seed=seed, outlier_frac=0.0, noise_px=noise)
    g = torch.Generator().manual_seed(seed + 1)
    out = torch.rand(B, N, generator=g) < outl
    if name.startswith("tail"):
        out[:, :2500] = True
    u = torch.where(out[..., None], torch.rand(B, N, 2, generator=g) * 64, b["pts2d"])
    counts = np.full(B, N, np.int32)
    if name.startswith("ragged"):
        counts = np.array([40, 3, 12, 25, 4, 40, 0, 33], np.int32)
    return dict(K=b["K"].numpy(), pts3d=b["pts3d"].numpy(), pts2d=u.numpy().astype(np.float32), counts=counts, reproj_err=np.float32(thr),
                iterations=np.int32(iters), seed=np.int32(rseed), pose_gt=b["pose"].numpy(), outlier=out.numpy())


def main():
    for name in (sys.argv[1:] or CASES):  # optionally only the named cases
        c = make_inputs(name)
        B, N = c["pts3d"].shape[:2]
        res = [O.ransac(c["K"][i], c["pts3d"][i], c["pts2d"][i], int(c["counts"][i]), float(c["reproj_err"]), int(c["iterations"]),
                        int(c["seed"]), i) for i in range(B)]
        mask, unsure = np.zeros((B, N), np.uint8), np.zeros((B, N), np.uint8)
        for i, r in enumerate(res):
            mask[i, r["inliers"]] = 1
            unsure[i, r.get("mask_unsure", [])] = 1  # points of the winner's mask within 1e-3 of the threshold: float32 may decide either way
        states = np.stack([np.concatenate((O.rot_to_quat(r["R"]), r["t"])) for r in res])
        np.savez_compressed(os.path.join(HERE, f"ransac_{name}.npz"), **{"in_" + k: v for k, v in c.items()},
                            best_hyp=np.array([r["best_hyp"] for r in res], np.int32), invalid=np.array([r["invalid"] for r in res], np.int32),
                            n_inliers=np.array([r["n_inliers"] for r in res], np.int32), inlier_mask=mask, mask_unsure=unsure, states=states,
                            per_hyp_count=np.stack([r.get("per_hyp_count", np.full(((int(c["iterations"]) + 63) // 64) * 64, -1)) for r in res]).astype(np.int16),
                            decided=np.array([r["decided"] for r in res]), mask_decided=np.array([r.get("mask_decided", r["decided"]) for r in res]))
        print(name, "decided", int(sum(r["decided"] for r in res)), "/", B, "invalid", int(sum(r["invalid"] for r in res)))


if __name__ == "__main__":
    main()
