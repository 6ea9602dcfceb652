"""The HIP weighted-PnP kernel (through the C ABI) against vectors of the real reference solve (Ceres 2.1.0), when
tests/golden/pnp_ceres_*.npz exist (see tests/golden/gen_golden_pnp_ceres.py; SKIPS until someone with a Ceres build runs it),
and against the true minimisers of the objective (fixture generated in the build container with SciPy/MINPACK)."""
import os

import numpy as np
import pytest
import torch

from oracle import pnp_oracle
from tests.pnp_cases import pnp_case, pose_err
from tests.test_oracle_pnp_ceres_golden import CERES, check_against_ceres, minimiser_distances

pytestmark = pytest.mark.gpu


def kernel_solve(c, **kw):
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(c[k]).to(dev) for k in ("K", "pts3d", "pts2d", "sqrtL", "start", "counts")}
    out = pnp_ceres.solve_device(t["K"], t["pts3d"], t["pts2d"], t["sqrtL"], t["start"], t["counts"], max_iter_count=c["max_iter"],
                                 function_tolerance=kw.get("ftol", c["ftol"]))
    return tuple(o.cpu().numpy() for o in out)


@pytest.mark.skipif(not CERES, reason="no tests/golden/pnp_ceres_*.npz (Ceres not buildable in the build image): PnP parity unpinned")
@pytest.mark.parametrize("path", CERES, ids=[os.path.basename(p) for p in CERES])
def test_kernel_vs_ceres_golden(path):
    check_against_ceres(path, kernel_solve)


def test_kernel_distance_to_true_minimiser():
    """Every accepted pose of the metric configuration against the fp64 MINPACK minimiser of the same objective; the kernel's
    distance distribution must be the oracle's (both stop by the same function-tolerance rule), job by job."""
    c = pnp_case("metric_B256_N64")
    st, tr, ret = kernel_solve(c)
    dq, dt = minimiser_distances(st, ret)
    so, tro, reto = pnp_oracle.solve_batched(c["start"], c["K"], c["pts2d"], c["pts3d"], c["sqrtL"], num_threads=4)
    dqo, dto = minimiser_distances(so, reto)
    print(f"kernel vs true minimiser: dq p50/p90/p99/max {np.quantile(dq, [.5, .9, .99, 1])}, dt {np.quantile(dt, [.5, .9, .99, 1])}")
    assert np.median(dq) < 2e-4 and np.median(dt) < 1e-4 and dq.max() < 2e-3 and dt.max() < 2e-3
    assert np.abs(dq - dqo).max() <= 1e-6 and np.abs(dt - dto).max() <= 1e-6, (np.abs(dq - dqo).max(), np.abs(dt - dto).max())
    # tight tolerance: the kernel lands on the minimisers themselves
    c2 = dict(c, max_iter=100)
    st, tr, ret = kernel_solve(c2, ftol=1e-13)
    dq, dt = minimiser_distances(st, ret)
    assert dq.max() < 2e-6 and dt.max() < 2e-6, (dq.max(), dt.max())
