#!/usr/bin/env python3
"""True minimisers of the weighted-PnP objective (ceres.cpp:15-65) for the metric configuration, computed in float64 by
SciPy's MINPACK Levenberg-Marquardt at machine-precision tolerances -- an optimiser that shares no code and no schedule
with Ceres, the oracle or the HIP kernel.  Any correct solver that stops by `function_tolerance = 1e-6` must end within
the stop rule's slack of these points; the GPU test reports that distance as a distribution.

    python tests/golden/gen_golden_pnp_minimiser.py        (build container; writes pnp_minimiser_metric_B256_N64.npz)
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.pnp_cases import pnp_case  # noqa: E402


def residual(x, K, X, u, L):
    R = Rotation.from_rotvec(x[:3]).as_matrix()
    p = X @ R.T + x[3:]
    k = K.reshape(-1)
    up = (p[:, 0] * k[0] + p[:, 1] * k[1]) / p[:, 2]
    vp = (p[:, 0] * k[3] + p[:, 1] * k[4]) / p[:, 2]
    du = up - (u[:, 0] - k[2])
    dv = vp - (u[:, 1] - k[5])
    return np.stack((du * L[:, 0, 0] + dv * L[:, 1, 0], dv * L[:, 1, 1]), -1).reshape(-1)


def main():
    name = "metric_B256_N64"
    c = pnp_case(name)
    B = len(c["start"])
    xs, costs = np.zeros((B, 7)), np.zeros(B)
    for i in range(B):
        st = c["start"][i].astype(np.float64)
        x0 = np.concatenate((Rotation.from_quat(np.roll(st[:4], -1)).as_rotvec(), st[4:]))
        args = tuple(a[i].astype(np.float64) for a in (c["K"], c["pts3d"], c["pts2d"], c["sqrtL"]))
        sol = least_squares(residual, x0, args=args, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
        sol = least_squares(residual, sol.x, args=args, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
        q = np.roll(Rotation.from_rotvec(sol.x[:3]).as_quat(), 1)
        xs[i] = np.concatenate((q if q[0] >= 0 else -q, sol.x[3:]))
        costs[i] = sol.cost
    out = os.path.join(HERE, f"pnp_minimiser_{name}.npz")
    np.savez_compressed(out, minimiser=xs, cost=costs)
    print(out, "max cost", costs.max())


if __name__ == "__main__":
    main()
