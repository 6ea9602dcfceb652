"""Trajectory-level lock-step of the HIP weighted-PnP kernel with the oracle: not only the final pose but the whole
trust-region schedule -- per iteration: accept / reject / invalid / which tolerance fired, cost, candidate cost, model cost
change, relative decrease, step norm, radius, gradient max-norm (what Ceres keeps in Solver::Summary::iterations) -- is
compared row by row.  A final-pose tolerance would hide a drifted schedule; this does not."""
import numpy as np
import pytest
import torch

from oracle import pnp_oracle
from tests.pnp_cases import pnp_case

pytestmark = pytest.mark.gpu
KIND, COST, CAND, MCC, RHO, STEP, RADIUS, GMAX = range(8)


def run_both(c, rows=50):
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(c[k]).to(dev) for k in ("K", "pts3d", "pts2d", "sqrtL", "start", "counts")}
    kw = dict(max_iter_count=c["max_iter"], function_tolerance=c["ftol"])
    st, tr, ret, it, trace = pnp_ceres.solve_device(t["K"], t["pts3d"], t["pts2d"], t["sqrtL"], t["start"], t["counts"], return_iters=True,
                                                    trace_rows=rows, **kw)
    st0, tr0, ret0, it0 = pnp_ceres.solve_device(t["K"], t["pts3d"], t["pts2d"], t["sqrtL"], t["start"], t["counts"], return_iters=True, **kw)
    # the diagnostic twin IS the shipped solve (same template body): bit-identical outputs
    assert torch.equal(st, st0) and torch.equal(tr, tr0) and torch.equal(ret, ret0) and torch.equal(it, it0)
    o = pnp_oracle.solve_batched_trace(c["start"], c["K"], c["pts2d"], c["pts3d"], c["sqrtL"], counts=c["counts"], max_iter=c["max_iter"],
                                       ftol=c["ftol"], num_threads=4, trace_rows=rows)
    return (st.cpu().numpy(), tr.cpu().numpy(), ret.cpu().numpy(), it.cpu().numpy(), trace.cpu().numpy()), o


def lockstep_mask(k_it, k_tr, o_it, o_tr):
    """jobs whose kind sequence (accept / reject / invalid / tolerance) is identical over the whole solve"""
    return (k_it == o_it) & (k_tr[:, :, KIND] == o_tr[:, :, KIND]).all(1)


def assert_rows_close(k_tr, o_tr, jobs):
    """Row-by-row agreement of the traced quantities.  Cost-like columns are compared on the scale of the job's cost at the start
    (a 3-point job converges to cost ~1e-20, where a relative difference means nothing); measured on the MI355X
    (scripts/pnp_numerics/lockstep_report.py -> profiles/r02/pnp_lockstep.txt): cost 3e-9, radius 7e-8, step norm 1e-5."""
    k, o = k_tr[jobs], o_tr[jobs]
    cost0 = np.maximum(o[:, :1, COST], 1e-30)
    gmax0 = np.maximum(o[:, :, GMAX].max(1, keepdims=True), 1e-30)
    fin = np.isfinite(o[:, :, CAND]) & (np.abs(o[:, :, CAND]) < 1e300)  # a failed candidate evaluation carries DBL_MAX in the oracle

    def close(col, rtol, scale, atol_rel, rows=None):
        a, b = k[:, :, col], o[:, :, col]
        m = fin if rows is None else fin & rows
        err = np.abs(a - b) - (rtol * np.abs(b) + atol_rel * scale)
        assert (err[m] <= 0).all(), (NAMES[col], float(np.max((np.abs(a - b) / np.maximum(np.abs(b), 1e-300))[m])))

    close(COST, 1e-8, cost0, 1e-8)
    close(CAND, 1e-8, cost0, 1e-8)
    close(MCC, 1e-4, cost0, 1e-8)
    close(STEP, 1e-4, 1.0, 1e-9)
    close(RADIUS, 1e-6, 1.0, 0.0)
    close(GMAX, 1e-3, gmax0, 1e-6)
    # the relative decrease steers the radius only on accepted / rejected steps (on a tolerance exit it is a ratio of two
    # differences at the 1e-6 level of the cost and carries their cancellation)
    decides = (o[:, :, KIND] == 1) | (o[:, :, KIND] == 2)
    close(RHO, 1e-3, 1.0, 1e-4, rows=decides & (np.abs(o[:, :, RHO]) < 10))


NAMES = ["kind", "cost", "cand_cost", "model_cost_change", "rho", "step_norm", "radius", "gmax"]


@pytest.mark.parametrize("name", ["metric_B256_N64", "dense_B16_N1024", "ragged_full_B64_N48", "identity_B64_N24", "maxiter1_B32_N64"])
def test_schedule_lockstep_well_posed(name):
    """Well-posed sets (incl. BASELINE configs[1]): EVERY job follows the oracle's schedule step for step; flags, iteration
    counts and radii identical."""
    c = pnp_case(name)
    (st, tr, ret, it, trace), (so, tro, reto, ito, traceo) = run_both(c)
    np.testing.assert_array_equal(ret, reto)
    np.testing.assert_array_equal(it, ito)
    same = lockstep_mask(it, trace, ito, traceo)
    assert same.all(), f"{name}: {int((~same).sum())} jobs left the oracle's accept/reject schedule"
    assert_rows_close(trace, traceo, np.arange(len(ret)))
    np.testing.assert_allclose(tr, tro, rtol=1e-6)


@pytest.mark.parametrize("name", ["hard_B512_N12", "minimal_B256_N4"])
def test_schedule_lockstep_hard(name):
    """Hard starts / near-minimal point sets: trajectories of up to 50 iterations through ill-conditioned steps amplify
    last-bit differences (two correct DENSE_QR implementations differ from each other at the same rate:
    profiles/r02/pnp_flip_rates.txt), so a few jobs leave the schedule.  Asserted: (1) >= 98 % of the jobs stay in lock-step
    for the WHOLE solve, and for those everything -- flags, radii, every traced quantity -- agrees; (2) a job that leaves the
    schedule was in agreement on every row before the split; (3) flags differ only among the jobs that left it."""
    c = pnp_case(name)
    (st, tr, ret, it, trace), (so, tro, reto, ito, traceo) = run_both(c)
    same = lockstep_mask(it, trace, ito, traceo)
    print(f"{name}: {int(same.sum())}/{len(same)} jobs in lock-step over the whole solve; flags differ in {int((ret != reto).sum())}")
    assert same.mean() >= 0.98
    np.testing.assert_array_equal(ret[same], reto[same])
    np.testing.assert_allclose(tr[same], tro[same], rtol=1e-6)
    assert_rows_close(trace, traceo, np.nonzero(same)[0])
    assert not (ret != reto)[same].any()
    for j in np.nonzero(~same)[0]:
        split = int(np.argmax(trace[j, :, KIND] != traceo[j, :, KIND])) if (trace[j, :, KIND] != traceo[j, :, KIND]).any() else min(it[j], ito[j])
        if split > 1:
            np.testing.assert_allclose(trace[j, :split - 1, RADIUS], traceo[j, :split - 1, RADIUS], rtol=1e-6)
