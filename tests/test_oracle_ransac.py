"""CPU checks of oracle/p3p_ransac_oracle.py (the checker of the GPU PnP initialiser): the hypothesis stream is a fixed integer
sequence, the independent Grunert P3P returns poses that satisfy the three-point problem, and the committed fixtures are what the
oracle produces from the seeded inputs."""
import glob
import os

import numpy as np

from oracle import p3p_ransac_oracle as O
from tests.golden.gen_golden_ransac import CASES, make_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hypothesis_stream_known_answers():
    # regression vectors of the integer stream (the GPU test proves the kernel draws the same one: best_hyp is compared exactly)
    assert O.hash_u32(0) == 0 and O.hash_u32(1) == 0x688990C0 and O.hash_u32(0xDEADBEEF) == 0xE628C683
    assert O.sample_indices(0, 0, 0, 64) == [39, 34, 62, 43] and O.sample_indices(7, 3, 191, 400) == [6, 336, 374, 23]
    a = [O.sample_indices(0, b, h, 64) for b in (0, 3) for h in (0, 1, 191)]
    assert all(len(set(i)) == 4 and max(i) < 64 for i in a)
    assert a[0] != a[1] and a[0] != a[3]
    small = O.sample_indices(5, 1, 7, 4)  # nl = 4: the four indices are a permutation
    assert sorted(small) == [0, 1, 2, 3]


def test_grunert_p3p_solves_the_three_point_problem():
    rng = np.random.default_rng(0)
    found = 0
    for _ in range(50):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        R *= np.sign(np.linalg.det(R))
        t = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(600, 1200)])
        x = rng.uniform(-40, 40, size=(3, 3))
        z = x @ R.T + t
        y = z / np.linalg.norm(z, axis=1, keepdims=True)
        sols = O.p3p_grunert(y, x)
        assert 1 <= len(sols) <= 4
        for Rs, ts in sols:  # every returned pose maps the model points onto their bearings with positive depth
            zz = x @ Rs.T + ts
            assert (zz[:, 2] > 0).all() and np.allclose(np.cross(zz, y), 0, atol=1e-7 * np.linalg.norm(zz)) and abs(np.linalg.det(Rs) - 1) < 1e-9
        found += any(np.allclose(Rs, R, atol=1e-6) and np.allclose(ts, t, atol=1e-4) for Rs, ts in sols)
    assert found == 50  # the true pose is always among them


def test_fixtures_are_the_oracle_on_the_seeded_inputs():
    files = sorted(glob.glob(os.path.join(GOLDEN, "ransac_*.npz")))
    assert len(files) == len(CASES)
    z = np.load(os.path.join(GOLDEN, "ransac_outliers_B24_N64.npz"))
    c = make_inputs("outliers_B24_N64")
    for k, v in c.items():
        np.testing.assert_array_equal(z["in_" + k], v)
    for i in (0, 11, 23):
        r = O.ransac(c["K"][i], c["pts3d"][i], c["pts2d"][i], int(c["counts"][i]), float(c["reproj_err"]), int(c["iterations"]), int(c["seed"]), i)
        assert r["best_hyp"] == z["best_hyp"][i] and r["n_inliers"] == z["n_inliers"][i]
        assert np.array_equal(np.nonzero(z["inlier_mask"][i])[0], r["inliers"])
    # the oracle's winner is a good pose: rotation within 0.05 of the ground truth on every pose of the noisy set with 25 % outliers
    q, gt = z["states"][:, :4], c["pose_gt"][:, :4]
    assert (np.abs(np.abs((q * gt).sum(1)) - 1) < 2e-3).all()
    # and its inlier sets reject the planted outliers
    planted = c["outlier"]
    assert (z["inlier_mask"].astype(bool) & planted).sum() <= 0.05 * planted.sum()
