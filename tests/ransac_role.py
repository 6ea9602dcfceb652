"""The ROLE contract of the PnP initialiser against vectors of the reference's `cv2_solver.solve` (tests/golden/gen_golden_ransac_cv2.py):
shared by the CPU (oracle) and GPU (kernel) tests and exercised against a stand-in by tests/test_golden_generators.py.

OpenCV's RANSAC cannot be matched hypothesis by hypothesis (its RNG, its EPnP).  What `test.py:59,120` needs from it, and what is checked:
  (1) validity: a pose the reference solves is solved here (the converse may differ on ill-posed rows: counted, bounded);
  (2) the same LM basin: the inlier refinement (`cer_solver.solve` with unit information on OUR inlier set, 20 iterations -- the step that
      follows in the chain) started from the reference's pose and from ours ends at the same pose within 1e-4;
  (3) the same consensus: IoU of the two inlier sets >= 0.9 on the noise-free set, >= 0.6 elsewhere (points within the threshold's reach of
      two different minimal-sample poses legitimately differ)."""
import glob
import os

import numpy as np

from tests.pnp_cases import pose_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CV2 = sorted(glob.glob(os.path.join(GOLDEN, "ransac_cv2_*.npz")))


def check_role(cv2_path, ransac, refine, golden_dir=GOLDEN):
    """ransac(K, X, U, counts, thr, iterations, seed) -> (states (B,7), inlier mask (B,N) bool, invalid (B) bool);
    refine(K, X, U, counts, mask, start) -> (states (B,7), rets (B))."""
    zc = np.load(cv2_path)
    name = os.path.basename(cv2_path)[len("ransac_cv2_"):-4]
    z = np.load(os.path.join(golden_dir, f"ransac_{name}.npz"))
    assert str(zc["problem_set"]) == f"ransac_{name}.npz"
    K, X, U, counts = z["in_K"], z["in_pts3d"], z["in_pts2d"], z["in_counts"]
    st, inl, bad = ransac(K, X, U, counts, float(z["in_reproj_err"]), int(z["in_iterations"]), int(z["in_seed"]))
    ref_bad, ref_st, ref_inl = zc["invalid"].astype(bool), zc["states"], zc["inlier_mask"].astype(bool)
    both = ~ref_bad & ~bad
    assert (bad & ~ref_bad).sum() <= 0.05 * len(bad) + 1, f"{name}: {int((bad & ~ref_bad).sum())} poses the reference solves are given up here"
    if not both.any():
        return dict(name=name, poses=0)
    rows = np.flatnonzero(both)
    ours, _ = refine(K[rows], X[rows], U[rows], counts[rows], inl[rows], st[rows].astype(np.float32))
    theirs, _ = refine(K[rows], X[rows], U[rows], counts[rows], inl[rows], ref_st[rows].astype(np.float32))
    dq, dt = pose_err(ours, theirs)
    inter, union = (inl[rows] & ref_inl[rows]).sum(1), (inl[rows] | ref_inl[rows]).sum(1)
    iou = inter / np.maximum(union, 1)
    floor = 0.9 if name.startswith("clean") else 0.6
    out = dict(name=name, poses=len(rows), dq=float(dq.max()), dt=float(dt.max()), iou_min=float(iou.min()))
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, out
    assert iou.min() >= floor, out
    return out
