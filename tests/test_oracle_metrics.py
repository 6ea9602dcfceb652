"""Pin oracle/pose_error_oracle.py against goldens from the reference's lib/utils/error6d.py."""
import glob
import os

import numpy as np

from oracle import pose_error_oracle as orc
from tests.util import GOLDEN


def test_pose_error_oracle_vs_reference():
    z = np.load(os.path.join(GOLDEN, "pose_err_b12_m700.npz"))
    pts = z["in_pts"].astype(np.float64)
    for i in range(len(z["in_R_est"])):
        e = orc.compute_pose_errors(z["in_R_est"][i].astype(np.float64), z["in_t_est"][i].astype(np.float64),
                                    z["in_R_gt"][i].astype(np.float64), z["in_t_gt"][i].astype(np.float64), pts)
        for k, v in e.items():
            assert abs(v - z["ref_" + k][i]) <= 1e-9 * max(1.0, abs(z["ref_" + k][i])), (i, k)
