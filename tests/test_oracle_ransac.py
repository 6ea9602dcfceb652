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


def test_fma32_is_an_exactly_rounded_float32_fma():
    """`fma32` against exact rational arithmetic, incl. sums constructed to sit on float32 midpoints (where a float64 intermediate double-rounds)."""
    from fractions import Fraction as F

    rng = np.random.default_rng(1)

    def exact(a, b, c):
        v = F(float(a)) * F(float(b)) + F(float(c))
        f = np.float32(float(v))
        cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
        return min(cands, key=lambda x: (abs(F(float(x)) - v), int(np.float32(x).view(np.uint32)) & 1))

    for trial in range(3000):
        a, b = (rng.standard_normal(2) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        if trial % 3 == 0:
            r = np.float32(rng.standard_normal() * 100)
            c = np.float32((float(r) + float(np.nextafter(r, np.float32(np.inf)))) / 2 - float(a) * float(b))
        else:
            c = np.float32(rng.standard_normal() * 10.0 ** rng.integers(-3, 4))
        got = O.fma32(np.array([a]), np.array([b]), np.array([c]))[0]
        assert got == exact(a, b, c), (a, b, c, got)


def test_float32_scoring_agrees_with_the_float64_scoring_away_from_the_threshold():
    """`score_f32` (the kernel's arithmetic) against a plain float64 evaluation of the same definition on the float64 oracle's own winner:
    the same inliers wherever a point is not within 1e-4 of the threshold, the error sum to 1e-5."""
    z = np.load(os.path.join(GOLDEN, "ransac_outliers_B24_N64.npz"))
    for i in (0, 5, 17):
        K, X, U, n = z["in_K"][i], z["in_pts3d"][i], z["in_pts2d"][i], int(z["in_counts"][i])
        r = O.ransac(K, X, U, n, float(z["in_reproj_err"]), int(z["in_iterations"]), int(z["in_seed"]), i)
        hyp = np.concatenate((r["R"].reshape(-1), r["t"])).astype(np.float32)[None]
        un, idet = O.normalised_points_f32(K, U[:n])
        thr2 = O.threshold2_f32(float(z["in_reproj_err"]), idet)
        inl, q = O.inlier_q_f32(hyp, X[:n], un, thr2)
        R, t = hyp[0, :9].astype(np.float64).reshape(3, 3), hyp[0, 9:].astype(np.float64)
        c = X[:n].astype(np.float64) @ R.T + t
        q64 = ((c[:, :2] - un.astype(np.float64) * c[:, 2:3]) ** 2).sum(1)
        lim = float(thr2) * c[:, 2] ** 2
        sure = np.abs(q64 - lim) > 1e-4 * lim
        assert np.array_equal(inl[0][sure], ((c[:, 2] > 0) & (q64 < lim))[sure]) and sure.sum() >= n - 2
        cnt, err = O.score_f32(hyp, X[:n], un, thr2)
        assert cnt[0] == inl[0].sum() and abs(float(err[0]) - q64[inl[0]].sum() / t[2] ** 2) <= 1e-5 * float(err[0]) + 1e-12


def test_float32_oracle_reproduces_the_kernel_records():
    """tests/golden/kernel_ransac_f32_*.npz are records of the HIP kernels (gen_golden_ransac_f32.py, taken on an MI355X): their float32 hypotheses, their
    per-hypothesis counts and inlier-error bits, their winners, inlier counts, masks and validity flags.  The float32-faithful oracle must
    reproduce every integer and every error bit from the hypotheses alone -- here, on the CPU."""
    files = sorted(glob.glob(os.path.join(GOLDEN, "kernel_ransac_f32_*.npz")))
    assert len(files) >= 3
    for path in files:
        k = np.load(path)
        z = np.load(os.path.join(GOLDEN, str(k["problem_set"])))
        K, X, U, counts = z["in_K"], z["in_pts3d"], z["in_pts2d"], z["in_counts"]
        for b in range(len(K)):
            r = O.ransac_f32(K[b], X[b], U[b], int(counts[b]), float(z["in_reproj_err"]), k["hyp32"][b])
            assert r["invalid"] == int(k["invalid"][b]) and r["best_hyp"] == int(k["best_hyp"][b]) and r["n_inliers"] == int(k["n_inliers"][b]), (path, b)
            assert np.array_equal(r["inlier_mask"], k["inlier_mask"][b].astype(bool)), (path, b)
            if r["per_hyp_count"] is not None:
                assert np.array_equal(r["per_hyp_count"], k["per_hyp_count"][b]), (path, b)
                assert np.array_equal(r["per_hyp_err"].view(np.uint32), k["per_hyp_err_bits"][b]), (path, b)
        # and the recorded hypotheses are P3P solutions of their samples: the winner against the independent float64 P3P
        for b in range(min(4, len(K))):
            if k["invalid"][b]:
                continue
            n = int(counts[b])
            un, _ = O.normalised_points_f32(K[b], U[b, :n])
            idx = O.sample_indices(int(z["in_seed"]), b, int(k["best_hyp"][b]), n)
            yb = np.concatenate((un[idx[:3]].astype(np.float64), np.ones((3, 1))), 1)
            yb /= np.linalg.norm(yb, axis=1, keepdims=True)
            Rk, tk = k["hyp64_winner"][b, :9].reshape(3, 3), k["hyp64_winner"][b, 9:]
            gap = min((np.abs(R - Rk).max() + np.abs(t - tk).max() / max(1.0, np.abs(tk).max()) for R, t in O.p3p_grunert(yb, X[b, idx[:3]].astype(np.float64))), default=np.inf)
            assert gap < 1e-6, (path, b, gap)
