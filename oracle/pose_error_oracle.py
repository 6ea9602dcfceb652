"""CPU oracle of the pose-error metrics (TEST INFRASTRUCTURE ONLY): numpy restatement of `lib/utils/error6d.py:87-154`
(add, adi, re, te) as bundled by `lib/utils/evaluate.py:333-339`.  Pinned by tests/golden/pose_err_*.npz generated from
the reference's error6d module (tests/test_oracle_metrics.py)."""
import math

import numpy as np
from scipy import spatial


def transform(pts, R, t):
    return pts @ R.T + t.reshape(1, 3)


def add(R_est, t_est, R_gt, t_gt, pts):
    return np.linalg.norm(transform(pts, R_est, t_est) - transform(pts, R_gt, t_gt), axis=1).mean()


def adi(R_est, t_est, R_gt, t_gt, pts):
    est, gt = transform(pts, R_est, t_est), transform(pts, R_gt, t_gt)
    d, _ = spatial.cKDTree(est).query(gt, k=1)
    return d.mean()


def re(R_est, R_gt):
    c = 0.5 * (np.trace(R_est @ np.linalg.inv(R_gt)) - 1.0)
    return math.degrees(math.acos(min(1.0, max(-1.0, float(c)))))


def te(t_est, t_gt):
    return float(np.linalg.norm(t_gt.reshape(3) - t_est.reshape(3)))


def compute_pose_errors(R_est, t_est, R_gt, t_gt, pts):
    return dict(adi=adi(R_est, t_est, R_gt, t_gt, pts), add=add(R_est, t_est, R_gt, t_gt, pts), re=re(R_est, R_gt), te=te(t_est, t_gt))
