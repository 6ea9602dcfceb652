"""The oracle (oracle/pnp_lm_oracle.c) against (a) vectors of the REAL reference solve (Ceres 2.1.0 behind ceres.cpp), when
tests/golden/pnp_ceres_*.npz exist -- they cannot be generated in the build image (no Ceres), see
tests/golden/gen_golden_pnp_ceres.py; until then these tests SKIP and PnP parity stays "unpinned" -- and (b) the true
minimisers of the same objective computed by an unrelated optimiser (SciPy/MINPACK, fp64, machine-precision tolerances)."""
import glob
import os

import numpy as np
import pytest

from oracle import pnp_oracle
from tests.pnp_cases import pnp_case, pose_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CERES = sorted(glob.glob(os.path.join(GOLDEN, "pnp_ceres_*.npz")))
# rets of two correct DENSE_QR trust-region solvers differ in this fraction of HARD jobs by round-off alone
# (profiles/r02/pnp_flip_rates.txt: same algorithm, column sums accumulated in reverse order); well-posed sets must be equal
RETS_SLACK = {"hard_B512_N12": 0.01, "minimal_B256_N4": 0.01}


def check_against_ceres(path, solve):
    z = np.load(path)
    name = os.path.basename(path)[len("pnp_ceres_"):-4]
    # the STORED inputs are the truth (the machine that ran Ceres may round the synthetic generator's transcendental functions
    # differently in the last bit); they must still be the seeded case, to fp32 rounding
    c = {k[3:]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("in_")}
    for k, v in pnp_case(name).items():
        np.testing.assert_allclose(np.asarray(c[k], np.float64), np.asarray(v, np.float64), rtol=1e-5, atol=1e-4,
                                   err_msg=f"{name}: input {k} is not the case of tests/pnp_cases.py")
    st, tr, ret = solve(c)
    ref_st, ref_tr, ref_ret = z["states"], z["result_tr"], z["rets"]
    flips = int((ret != ref_ret).sum())
    assert flips <= RETS_SLACK.get(name, 0.0) * len(ret), f"{name}: rets differ from Ceres in {flips} of {len(ret)} jobs"
    same = ret == ref_ret
    bad = same & (ref_ret == 1)
    np.testing.assert_array_equal(st[bad], c["start"][bad])  # invalid -> untouched (ceres.cpp:134-138)
    ok = same & (ref_ret == 0)
    if not ok.any():  # e.g. max_iter = 1: every job ends NO_CONVERGENCE -- flags, radii and the untouched states were the whole check
        np.testing.assert_allclose(tr[same], ref_tr[same], rtol=1e-3)
        print(f"{name}: {flips} flag flips; no job accepted by both")
        return
    dq, dt = pose_err(st[ok], ref_st[ok])
    print(f"{name}: {flips} flag flips; dq p50/p99/max {np.median(dq):.1e}/{np.quantile(dq, .99):.1e}/{dq.max():.1e}, "
          f"dt {np.median(dt):.1e}/{np.quantile(dt, .99):.1e}/{dt.max():.1e}")
    far = (dq > 1e-4) | (dt > 1e-4)
    assert far.sum() <= RETS_SLACK.get(name, 0.0) * len(ret), f"{name}: {int(far.sum())} accepted poses further than 1e-4 from Ceres"
    np.testing.assert_allclose(tr[same], ref_tr[same], rtol=1e-3)


def oracle_solve(c):
    return pnp_oracle.solve_batched(c["start"], c["K"], c["pts2d"], c["pts3d"], c["sqrtL"], counts=c["counts"],
                                    max_iter=c["max_iter"], ftol=c["ftol"], num_threads=4)


@pytest.mark.skipif(not CERES, reason="no tests/golden/pnp_ceres_*.npz: Ceres 2.1.0 is not buildable in this image -- run "
                                      "tests/golden/gen_golden_pnp_ceres.py where the reference's extension exists (PnP parity unpinned)")
@pytest.mark.parametrize("path", CERES, ids=[os.path.basename(p) for p in CERES])
def test_oracle_vs_ceres_golden(path):
    check_against_ceres(path, oracle_solve)


def minimiser_distances(states, ret):
    z = np.load(os.path.join(GOLDEN, "pnp_minimiser_metric_B256_N64.npz"))
    assert int(ret.sum()) == 0
    return pose_err(states.astype(np.float64), z["minimiser"])


def test_oracle_distance_to_true_minimiser():
    """function_tolerance = 1e-6 ends the trust-region loop at the first step whose cost change is below 1e-6 of the cost --
    and that last candidate is NOT accepted (Ceres' FunctionToleranceReached returns before the step is taken) -- so the
    returned pose is one step short of the minimum: measured here median 8e-5 / max 8e-4 in q.  (The 1e-4 tolerance of the
    north star is agreement with the reference's result, which stops at the same point, not distance to the minimum.)"""
    c = pnp_case("metric_B256_N64")
    st, tr, ret = oracle_solve(c)
    dq, dt = minimiser_distances(st, ret)
    print(f"oracle vs true minimiser: dq p50/p90/p99/max {np.quantile(dq, [.5, .9, .99, 1])}, dt {np.quantile(dt, [.5, .9, .99, 1])}")
    assert np.median(dq) < 2e-4 and np.median(dt) < 1e-4
    assert dq.max() < 2e-3 and dt.max() < 2e-3
    # with a tight function tolerance the same loop lands ON the minimisers (fp32 output rounding: 6e-8 in q, 6e-8 rel in t)
    st, tr, ret = pnp_oracle.solve_batched(c["start"], c["K"], c["pts2d"], c["pts3d"], c["sqrtL"], ftol=1e-13, max_iter=100, num_threads=4)
    dq, dt = minimiser_distances(st, ret)
    assert dq.max() < 2e-6 and dt.max() < 2e-6, (dq.max(), dt.max())
