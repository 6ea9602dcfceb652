"""Pin of the oracle's OPTIMISER (oracle/pnp_lm_oracle.c: lm_minimize -- trust-region loop, Levenberg-Marquardt damping, Jacobi
scaling, dense QR step, step acceptance, radius schedule, termination tests) against output PUBLISHED by Ceres Solver itself:
the per-iteration log of Powell's function in the Ceres tutorial (tests/golden/ceres_powell_published.txt).  Ceres cannot be
built in this image and the reference holds no vector at its Ceres boundary, so this is the one place where numbers produced
by the real dependency are available; the PnP solve runs through the same lm_minimize, differing only in the residual function
(whose values and Jacobian are checked separately: SciPy/MINPACK minimisers, tests/test_oracle_pnp.py).

Every cell of the published table is compared as a STRING at the printed precision (%.6e for the cost, %.2e elsewhere)."""
import os

import numpy as np

from oracle import pnp_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KIND, COST, CAND, MCC, RHO, STEP, RADIUS, GMAX = range(8)


def published(name="ceres_powell_published.txt"):
    rows, notes = [], []
    for ln in open(os.path.join(HERE, "golden", name)):
        if ln.startswith("#"):
            notes.append(ln[1:].strip())
        elif ln.strip():
            rows.append(ln.split())
    return rows, notes


def test_minimiser_reproduces_the_published_ceres_log_of_powells_function():
    rows, notes = published()
    ok, x, iters, radius, (cost0, g0), trace = pnp_oracle.powell_trace()
    assert ok and iters == len(rows) - 1 == 14
    assert [f"{0:d}", f"{cost0:.6e}", "0.00e+00", f"{g0:.2e}", "0.00e+00", "0.00e+00", "1.00e+04"] == rows[0]
    for i, row in enumerate(rows[1:]):
        t = trace[i]
        assert int(t[KIND]) == 1  # every step of the published run is a successful one
        mine = [f"{i + 1:d}", f"{t[CAND]:.6e}", f"{t[COST] - t[CAND]:.2e}", f"{t[GMAX]:.2e}", f"{t[STEP]:.2e}", f"{t[RHO]:.2e}", f"{t[RADIUS]:.2e}"]
        assert mine == row, (mine, row)
    # Termination: CONVERGENCE (Gradient tolerance reached. Gradient max norm: 3.642190e-11 <= 1.000000e-10)
    term = next(n for n in notes if n.startswith("Termination"))
    assert f"Gradient max norm: {trace[-1][GMAX]:.6e} <= 1.000000e-10" in term
    # Final x1 = 0.000146222, x2 = -1.46222e-05, x3 = 2.40957e-05, x4 = 2.40957e-05   (printed with operator<<: 6 significant digits)
    final = next(n for n in notes if n.startswith("Final"))
    assert final == "Final " + ", ".join(f"x{j + 1} = {x[j]:.6g}" for j in range(4))
    assert abs(radius - 4.78e10) < 0.005e10


def test_published_log_is_sensitive_to_the_schedule():
    """The comparison is not vacuous: each ingredient of the restated schedule moves printed digits of the log -- a different
    initial radius, function tolerance or iteration cap changes rows or the termination."""
    _, _, iters_cap, _, _, _ = pnp_oracle.powell_trace(max_iter=5)
    assert iters_cap == 5
    ok, x, iters, radius, _, trace = pnp_oracle.powell_trace(ftol=0.99)  # |dcost| <= 0.99 cost fires at the very first step
    assert ok and iters == 1 and int(trace[0][KIND]) == 4
    assert np.allclose(x, [3.0, -1.0, 0.0, 1.0])  # and Ceres does not take that step (FunctionToleranceReached returns first)


def test_minimiser_reproduces_the_published_ceres_log_of_hello_world():
    """f = 10 - x from x = 0.5: the cost after the first step, 4.511598e-07, is what the Levenberg-Marquardt damping at radius 1e4
    leaves of 45.125; two successful steps are logged, the third iteration ends the run by the parameter tolerance (so the report
    says "Iterations: 2 ... Final cost: 5.012552e-16") and x prints as 10."""
    rows, notes = published("ceres_helloworld_published.txt")
    ok, x, iters, radius, trace = pnp_oracle.hello_trace()
    assert ok and iters == 3 and int(trace[2][KIND]) == 3  # ParameterToleranceReached in iteration 3
    assert [f"{0.5 * 9.5 ** 2:.6e}", "9.50e+00"] == [rows[0][1], rows[0][3]]
    for i, row in enumerate(rows[1:]):
        t = trace[i]
        assert int(t[KIND]) == 1
        mine = [f"{i + 1:d}", f"{t[CAND]:.6e}", f"{t[COST] - t[CAND]:.2e}", f"{t[GMAX]:.2e}", f"{t[STEP]:.2e}", f"{t[RHO]:.2e}", f"{t[RADIUS]:.2e}"]
        assert mine == row, (mine, row)
    assert f"{x:.6g}" == "10" and f"Final cost: {trace[1][CAND]:.6e}" in " ".join(notes)


def test_radius_schedule_reproduces_the_published_curve_fitting_log():
    """The third tutorial example (curve fitting, examples/curve_fitting.cc) starts with FIVE rejected steps; its published log
    prints, per iteration, the step quality `tr_ratio` and the radius that follows.  The residual data of that example are not
    restated here, but the radius column is a pure function of the tr_ratio column and of the schedule -- initial radius 1e4,
    rejected: radius /= 2, 4, 8, 16, 32 (the divisor doubles, reset by a success), accepted: radius /= max(1/3, 1 - (2 rho - 1)^3) --
    and the oracle's own update functions reproduce every printed radius from the printed ratios."""
    tr_ratio = [-1.87e+01, -1.86e+01, -1.85e+01, -1.70e+01, -6.32e+00, 1.37e+00, 1.10e+00, 1.03e+00, 9.94e-01, 9.89e-01, 9.97e-01,
                1.00e+00, 1.00e+00]
    tr_radius = ["5.00e+03", "1.25e+03", "1.56e+02", "9.77e+00", "3.05e-01", "9.16e-01", "2.75e+00", "8.24e+00", "2.47e+01", "7.42e+01",
                 "2.22e+02", "6.67e+02", "2.00e+03"]
    got = [f"{r:.2e}" for r in pnp_oracle.radius_schedule(tr_ratio)]
    assert got == tr_radius, got
