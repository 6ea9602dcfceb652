"""GPU parity of the batched pose-error kernel against goldens from the reference's error6d.py and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def test_pose_errors_vs_reference_golden():
    from lc_amd.metrics import compute_pose_errors

    z = np.load(os.path.join(GOLDEN, "pose_err_b12_m700.npz"))
    dev = torch.device("cuda:0")
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    e = compute_pose_errors(t("in_R_est"), t("in_t_est"), t("in_R_gt"), t("in_t_gt"), t("in_pts"))
    for k in ("adi", "add", "te"):
        ref = z["ref_" + k]
        assert np.abs(e[k].cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
    # re: acos near 0 amplifies the fp32 rounding of R (1e-7 in the trace -> ~0.03 deg); compare away from 0 tightly
    ref = z["ref_re"]
    got = e["re"].cpu().numpy()
    assert np.abs(got - ref)[ref > 0.5].max() <= 1e-3 and np.abs(got - ref).max() <= 0.05


def test_pose_errors_packed_objects_and_large_cloud():
    """Two objects with different vertex counts packed back to back (per-pose offset/count), M > one LDS tile."""
    from lc_amd.metrics import compute_pose_errors
    from oracle import pose_error_oracle as orc
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(3)
    cnt = [1500, 2300]
    pts = (rng.random((sum(cnt), 3)).astype(np.float32) * 2 - 1) * 40
    B = 6
    obj = np.array([0, 1, 1, 0, 1, 0])
    off = np.array([0, cnt[0]])[obj].astype(np.int32)
    num = np.array(cnt)[obj].astype(np.int32)
    Rg = Rotation.random(B, random_state=4).as_matrix().astype(np.float32)
    Re = (Rg @ Rotation.from_rotvec(rng.normal(size=(B, 3)) * 0.1).as_matrix()).astype(np.float32)
    tg = rng.normal(size=(B, 3)).astype(np.float32) * 30 + np.array([0, 0, 800], np.float32)
    te = tg + rng.normal(size=(B, 3)).astype(np.float32) * 5
    dev = torch.device("cuda:0")
    e = compute_pose_errors(*(torch.from_numpy(a).to(dev) for a in (Re, te, Rg, tg, pts)), pts_off=torch.from_numpy(off),
                            pts_cnt=torch.from_numpy(num))
    for i in range(B):
        p = pts[off[i]:off[i] + num[i]].astype(np.float64)
        r = orc.compute_pose_errors(Re[i].astype(np.float64), te[i].astype(np.float64), Rg[i].astype(np.float64), tg[i].astype(np.float64), p)
        for k in ("adi", "add", "te"):
            assert abs(float(e[k][i]) - r[k]) <= 2e-5 * max(1.0, r[k]), (i, k)
        assert abs(float(e["re"][i]) - r["re"]) <= 1e-3
