/*
 * CPU oracle for the weighted PnP solve (TEST INFRASTRUCTURE ONLY -- never the product path).
 *
 * Restates /root/reference/lib/pnp/cxx/ceres.cpp:15-177 in plain C99 + OpenMP and exports the same
 * C ABI (`lib/pnp/cxx/ext.h:2-15`).  The optimiser itself lives in the third-party dependency
 * Ceres Solver 2.1.0 (pinned by `scripts/build-ceres.sh:18-22`), whose source is NOT under
 * /root/reference and cannot be built in this image (no Eigen/glog/SuiteSparse, no network), so its
 * published default algorithm is restated here from the call site's options (ceres.cpp:118-125:
 * DENSE_QR, max_num_iterations, function_tolerance; everything else default):
 *
 *   trust-region Levenberg-Marquardt, initial radius 1e4, max radius 1e16, min radius 1e-32,
 *   min_relative_decrease 1e-3, Jacobi column scaling 1/(1+||J_j||) fixed at iteration 0,
 *   LM diagonal = sqrt(clamp(diag(Js^T Js),1e-6,1e32)/radius), step = argmin ||Js y + r||^2 + ||D y||^2
 *   by dense QR of [Js; D], radius /= max(1/3, 1-(2 rho-1)^3) on success, /= 2,4,8.. on failure,
 *   stop: |dcost| <= ftol*cost (CONVERGENCE), ||dx|| <= 1e-8(||x||+1e-8) (CONVERGENCE),
 *   max|g| <= 1e-10 (CONVERGENCE), iteration >= max_num_iterations (NO_CONVERGENCE -> invalid),
 *   5 consecutive invalid steps (FAILURE -> invalid).
 *
 * PARITY STATUS.  No vector produced by the reference's own pnp_ceres_f32 exists (Ceres is not buildable here, the
 * reference holds no test at this boundary: SURVEY.md 8c), so the COMPOSITE is unpinned in the strict sense; its two halves
 * are pinned separately:
 *   - the optimiser (lm_minimize below: the loop every PnP solve runs through) reproduces, to every printed digit, the
 *     per-iteration logs that the Ceres documentation publishes for its tutorial problems -- Powell's function (14 iterations:
 *     cost, cost change, |gradient|, |step|, trust-region ratio and radius, termination by the gradient tolerance, final x)
 *     and hello-world (the LM damping at radius 1e4, exit by the parameter tolerance): tests/golden/ceres_*_published.txt,
 *     tests/test_oracle_ceres_published.py;
 *   - the residual model (ceres.cpp:15-65) by optimisers that share nothing with this file (SciPy/MINPACK minimisers,
 *     tests/test_oracle_pnp.py, tests/golden/pnp_minimiser_*.npz) and known-answer tests (noise-free recovery, stationarity,
 *     <3 points, full 2x2 icov);
 *   - tests/golden/gen_golden_pnp_ceres.py produces reference vectors of the composite wherever the reference's extension exists.
 *
 * Residual model (ceres.cpp:15-65): p = R(aa) X + t; up = (p0 k0 + p1 k1)/p2, vp = (p0 k3 + p1 k4)/p2,
 * du = up-(u-k2), dv = vp-(v-k5); r = [du*L00 + dv*L10, dv*L11]; no z clamp; doubles inside.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ceres/rotation.h QuaternionToAngleAxis (ceres.cpp:96) */
static void quat_to_aa(const double q[4], double aa[3]) {
    const double q1 = q[1], q2 = q[2], q3 = q[3];
    const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
    if (s2 > 0.0) {
        const double s = sqrt(s2), c = q[0];
        const double two_theta = 2.0 * ((c < 0.0) ? atan2(-s, -c) : atan2(s, c));
        const double k = two_theta / s;
        aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
    } else {
        aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0;
    }
}

/* ceres/rotation.h AngleAxisToQuaternion (ceres.cpp:131) */
static void aa_to_quat(const double aa[3], double q[4]) {
    const double t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (t2 > 0.0) {
        const double t = sqrt(t2), h = t * 0.5, k = sin(h) / t;
        q[0] = cos(h); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
    } else {
        q[0] = 1.0; q[1] = aa[0] * 0.5; q[2] = aa[1] * 0.5; q[3] = aa[2] * 0.5;
    }
}

typedef struct {
    int n;
    const float *u, *X, *L;
    double cam[6];
} problem_t;

/* residuals (2n) and, if J != NULL, the 2n x 6 row-major Jacobian w.r.t. (aa, t).
 * AngleAxisRotatePoint (ceres.cpp:37) incl. its small-angle branch; derivative = exact derivative of the
 * same branch (what the Jet autodiff of ceres.cpp:59 produces). returns 0 if any value is non-finite. */
static int evaluate(const problem_t *P, const double x[6], double *r, double *J, double *cost) {
    const double *aa = x, *t = x + 3;
    const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    const int big = theta2 > DBL_EPSILON;
    double th = 0, c = 1, s = 0, w[3] = {0, 0, 0};
    if (big) {
        th = sqrt(theta2); c = cos(th); s = sin(th);
        w[0] = aa[0] / th; w[1] = aa[1] / th; w[2] = aa[2] / th;
    }
    const double *k = P->cam;
    double sum = 0;
    int ok = 1;
    for (int i = 0; i < P->n; ++i) {
        const double p[3] = {P->X[3 * i], P->X[3 * i + 1], P->X[3 * i + 2]};
        double f[3], D[3][3];
        if (big) {
            const double wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
            const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
            const double tmp = wp * (1.0 - c);
            for (int a = 0; a < 3; ++a) f[a] = p[a] * c + wxp[a] * s + w[a] * tmp;
            if (J) {
                /* d f/d theta (x) w^T + d f/d w * (I - w w^T)/theta */
                double dth[3], M[3][3];
                for (int a = 0; a < 3; ++a) dth[a] = -p[a] * s + wxp[a] * c + w[a] * wp * s;
                /* M = -[p]x s + (wp I + w p^T)(1-c) */
                const double px[3][3] = {{0, -p[2], p[1]}, {p[2], 0, -p[0]}, {-p[1], p[0], 0}};
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        M[a][b] = -px[a][b] * s + ((a == b ? wp : 0.0) + w[a] * p[b]) * (1.0 - c);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        double acc = 0;
                        for (int d = 0; d < 3; ++d) acc += M[a][d] * ((d == b ? 1.0 : 0.0) - w[d] * w[b]);
                        D[a][b] = dth[a] * w[b] + acc / th;
                    }
            }
        } else {
            f[0] = p[0] + aa[1] * p[2] - aa[2] * p[1];
            f[1] = p[1] + aa[2] * p[0] - aa[0] * p[2];
            f[2] = p[2] + aa[0] * p[1] - aa[1] * p[0];
            if (J) {
                const double npx[3][3] = {{0, p[2], -p[1]}, {-p[2], 0, p[0]}, {p[1], -p[0], 0}};
                memcpy(D, npx, sizeof(D));
            }
        }
        const double q0 = f[0] + t[0], q1 = f[1] + t[1], q2 = f[2] + t[2];
        const double nu = q0 * k[0] + q1 * k[1], nv = q0 * k[3] + q1 * k[4];
        const double up = nu / q2, vp = nv / q2;
        const double du = up - ((double)P->u[2 * i] - k[2]);
        const double dv = vp - ((double)P->u[2 * i + 1] - k[5]);
        const double a = P->L[4 * i], b = P->L[4 * i + 2], cc = P->L[4 * i + 3];
        const double r0 = du * a + dv * b, r1 = dv * cc;
        r[2 * i] = r0; r[2 * i + 1] = r1;
        sum += r0 * r0 + r1 * r1;
        if (!isfinite(r0) || !isfinite(r1)) ok = 0;
        if (J) {
            const double dup[3] = {k[0] / q2, k[1] / q2, -nu / (q2 * q2)};
            const double dvp[3] = {k[3] / q2, k[4] / q2, -nv / (q2 * q2)};
            double d0[3], d1[3];
            for (int m = 0; m < 3; ++m) { d0[m] = a * dup[m] + b * dvp[m]; d1[m] = cc * dvp[m]; }
            double *J0 = J + 12 * i, *J1 = J0 + 6;
            for (int m = 0; m < 3; ++m) {
                J0[m] = d0[0] * D[0][m] + d0[1] * D[1][m] + d0[2] * D[2][m];
                J1[m] = d1[0] * D[0][m] + d1[1] * D[1][m] + d1[2] * D[2][m];
                J0[3 + m] = d0[m];
                J1[3 + m] = d1[m];
                if (!isfinite(J0[m]) || !isfinite(J1[m]) || !isfinite(d0[m]) || !isfinite(d1[m])) ok = 0;
            }
        }
    }
    *cost = 0.5 * sum;
    return ok;
}

/* least squares  min ||A y - b||  for A (m x 6, row-major, overwritten), b (m, overwritten) by Householder QR.
 * returns 0 on rank deficiency / non-finite. (DENSE_QR, ceres.cpp:119) */
static int qr_solve6(double *A, double *b, int m, double y[6]) {
    for (int j = 0; j < 6; ++j) {
        double nrm = 0;
        for (int i = j; i < m; ++i) nrm += A[i * 6 + j] * A[i * 6 + j];
        nrm = sqrt(nrm);
        if (!(nrm > 0) || !isfinite(nrm)) return 0;
        const double alpha = A[j * 6 + j] > 0 ? -nrm : nrm;
        const double v0 = A[j * 6 + j] - alpha;
        double vnorm2 = v0 * v0;
        for (int i = j + 1; i < m; ++i) vnorm2 += A[i * 6 + j] * A[i * 6 + j];
        if (vnorm2 > 0) {
            for (int c = j + 1; c < 6; ++c) {
                double d = v0 * A[j * 6 + c];
                for (int i = j + 1; i < m; ++i) d += A[i * 6 + j] * A[i * 6 + c];
                d = 2 * d / vnorm2;
                A[j * 6 + c] -= d * v0;
                for (int i = j + 1; i < m; ++i) A[i * 6 + c] -= d * A[i * 6 + j];
            }
            double d = v0 * b[j];
            for (int i = j + 1; i < m; ++i) d += A[i * 6 + j] * b[i];
            d = 2 * d / vnorm2;
            b[j] -= d * v0;
            for (int i = j + 1; i < m; ++i) b[i] -= d * A[i * 6 + j];
        }
        A[j * 6 + j] = alpha;
    }
    for (int j = 5; j >= 0; --j) {
        double acc = b[j];
        for (int c = j + 1; c < 6; ++c) acc -= A[j * 6 + c] * y[c];
        y[j] = acc / A[j * 6 + j];
        if (!isfinite(y[j])) return 0;
    }
    return 1;
}

/* The linear solver of the LM step.  DENSE_QR (ceres.cpp:119) is the restatement; scripts/pnp_numerics/ re-includes this
 * file with another PNP_STEP_SOLVER to measure how far other factorisations drift from it (never used by the tests). */
#ifndef PNP_STEP_SOLVER
#define PNP_STEP_SOLVER qr_solve6
#endif

/* Optional per-iteration trace (tests/test_gpu_pnp_trace.py locks the kernel's schedule to it step by step).
 * One row of PNP_TRACE_COLS doubles per trust-region iteration:
 *   0 kind (0 invalid step, 1 accepted, 2 rejected, 3 parameter tolerance, 4 function tolerance)
 *   1 cost at x before the step   2 candidate cost   3 model cost change   4 relative decrease rho
 *   5 ||delta|| (unscaled step)   6 radius AFTER the iteration's update    7 max|gradient| after the iteration */
#define PNP_TRACE_COLS 8
enum { TR_INVALID = 0, TR_ACCEPT = 1, TR_REJECT = 2, TR_PTOL = 3, TR_FTOL = 4 };

/* The minimiser itself -- Ceres 2.1.0's TrustRegionMinimizer with the LevenbergMarquardtStrategy and DENSE_QR, as restated in
 * the header -- over ANY residual function of six parameters (fewer: pad with parameters that no residual depends on; their
 * Jacobian columns are zero, the damping rows keep [J; D] full rank and their step is exactly zero).  The PnP solve below and the
 * published-trace check (oracle_powell_trace: Powell's function from the Ceres tutorial) both run through it.
 *   eval(ctx, x, r, J or NULL, &cost) -> 0 if anything is non-finite;  m residuals;  x in/out.
 *   returns 1 when the termination type is CONVERGENCE (gradient / parameter / function tolerance or minimum radius), else 0. */
typedef int (*eval_fn)(const void *ctx, const double x[6], double *r, double *J, double *cost);

/* LevenbergMarquardtStrategy::StepAccepted / StepRejected (StepIsInvalid == StepRejected(0)): the radius schedule */
static void radius_step_accepted(double *radius, double *decrease_factor, double step_quality) {
    const double tq = 2.0 * step_quality - 1.0;
    *radius = *radius / fmax(1.0 / 3.0, 1.0 - tq * tq * tq);
    *radius = fmin(1e16, *radius); /* max_trust_region_radius */
    *decrease_factor = 2.0;
}
static void radius_step_rejected(double *radius, double *decrease_factor) {
    *radius = *radius / *decrease_factor;
    *decrease_factor *= 2.0;
}
/* test hook: the schedule driven by a given accept (quality > 0) / reject (quality <= 0) sequence */
void oracle_radius_schedule(const double *quality, int n, double *radii) {
    double radius = 1e4, factor = 2.0;
    for (int i = 0; i < n; ++i) {
        if (quality[i] > 1e-3) radius_step_accepted(&radius, &factor, quality[i]);
        else radius_step_rejected(&radius, &factor);
        radii[i] = radius;
    }
}

static int lm_minimize(eval_fn evaluate_fn, const void *ctx, int m, double x[6], int maxIterCnt, double ftol, int printSummary,
                       double *radius_out, double *trace, int trace_rows, int *n_iter) {
#define TRACE(kind, cc, mcc, rho, sn)                                                          \
    do {                                                                                       \
        if (trace && iter <= trace_rows) {                                                     \
            double *row_ = trace + (size_t)(iter - 1) * PNP_TRACE_COLS;                        \
            row_[0] = (kind); row_[1] = x_cost; row_[2] = (cc); row_[3] = (mcc); row_[4] = (rho); \
            row_[5] = (sn); row_[6] = radius; row_[7] = gmax;                                  \
        }                                                                                      \
    } while (0)
    double *r = (double *)malloc(sizeof(double) * (size_t)(m + 6) * 2);
    double *rc = r + (m + 6);
    double *J = (double *)malloc(sizeof(double) * (size_t)(m + 6) * 6 * 2);
    double *Aw = J + (size_t)(m + 6) * 6;

    const double ptol = 1e-8, gtol = 1e-10;
    const double min_rel_decrease = 1e-3, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    double x_cost = 0, scale[6], g[6], gmax = 0, x_norm = 0;
    int converged = 0, failed = 0, iter = 0, n_invalid = 0;

    if (!evaluate_fn(ctx, x, r, J, &x_cost)) failed = 1; /* "Initial residual and Jacobian evaluation failed." */
    if (!failed) {
        for (int j = 0; j < 6; ++j) {
            double cn = 0;
            g[j] = 0;
            for (int i = 0; i < m; ++i) { cn += J[i * 6 + j] * J[i * 6 + j]; g[j] += J[i * 6 + j] * r[i]; }
            scale[j] = 1.0 / (1.0 + sqrt(cn));
        }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < 6; ++j) J[i * 6 + j] *= scale[j];
        gmax = 0; x_norm = 0;
        for (int j = 0; j < 6; ++j) { gmax = fmax(gmax, fabs(g[j])); x_norm += x[j] * x[j]; }
        x_norm = sqrt(x_norm);
    }
    while (!failed && !converged) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (iter >= maxIterCnt) break;                 /* NO_CONVERGENCE */
        if (gmax <= gtol) { converged = 1; break; }    /* gradient tolerance */
        if (radius <= min_radius) { converged = 1; break; }
        ++iter;
        /* LevenbergMarquardtStrategy::ComputeStep */
        double y[6], step[6], delta[6], xc[6];
        for (int j = 0; j < 6; ++j) {
            double cn = 0;
            for (int i = 0; i < m; ++i) cn += J[i * 6 + j] * J[i * 6 + j];
            cn = fmin(fmax(cn, 1e-6), 1e32);
            const double d = sqrt(cn / radius);
            for (int c = 0; c < 6; ++c) Aw[(size_t)(m + j) * 6 + c] = (c == j) ? d : 0.0;
            rc[m + j] = 0;
        }
        memcpy(Aw, J, sizeof(double) * (size_t)m * 6);
        memcpy(rc, r, sizeof(double) * (size_t)m);
        int step_ok = PNP_STEP_SOLVER(Aw, rc, m + 6, y);
        double model_cost_change = 0;
        if (step_ok) {
            for (int j = 0; j < 6; ++j) step[j] = -y[j];
            for (int i = 0; i < m; ++i) {
                double mr = 0;
                for (int j = 0; j < 6; ++j) mr += J[i * 6 + j] * step[j];
                model_cost_change -= mr * (r[i] + mr / 2.0);
            }
            step_ok = model_cost_change > 0.0;
        }
        if (!step_ok) { /* HandleInvalidStep */
            if (++n_invalid >= 5) { failed = 1; TRACE(TR_INVALID, 0.0, model_cost_change, 0.0, 0.0); break; }
            radius_step_rejected(&radius, &decrease_factor);
            TRACE(TR_INVALID, 0.0, model_cost_change, 0.0, 0.0);
            continue;
        }
        n_invalid = 0;
        double step_norm = 0;
        for (int j = 0; j < 6; ++j) { delta[j] = step[j] * scale[j]; xc[j] = x[j] + delta[j]; step_norm += delta[j] * delta[j]; }
        step_norm = sqrt(step_norm);
        double cand_cost;
        if (!evaluate_fn(ctx, xc, rc, NULL, &cand_cost)) cand_cost = DBL_MAX;
        if (step_norm <= ptol * (x_norm + ptol)) {                                   /* ParameterToleranceReached */
            converged = 1; TRACE(TR_PTOL, cand_cost, model_cost_change, 0.0, step_norm); break;
        }
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= ftol * x_cost) {                                    /* FunctionToleranceReached */
            converged = 1; TRACE(TR_FTOL, cand_cost, model_cost_change, cost_change / model_cost_change, step_norm); break;
        }
        const double rel = cost_change / model_cost_change;
        if (rel > min_rel_decrease) { /* HandleSuccessfulStep */
            const double prev_cost = x_cost;
            memcpy(x, xc, sizeof(xc));
            x_norm = 0;
            for (int j = 0; j < 6; ++j) x_norm += x[j] * x[j];
            x_norm = sqrt(x_norm);
            if (!evaluate_fn(ctx, x, r, J, &x_cost)) { failed = 1; break; }
            gmax = 0;
            for (int j = 0; j < 6; ++j) {
                g[j] = 0;
                for (int i = 0; i < m; ++i) g[j] += J[i * 6 + j] * r[i];
                gmax = fmax(gmax, fabs(g[j]));
            }
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < 6; ++j) J[i * 6 + j] *= scale[j];
            radius_step_accepted(&radius, &decrease_factor, rel);
            { const double now_cost = x_cost; x_cost = prev_cost; TRACE(TR_ACCEPT, cand_cost, model_cost_change, rel, step_norm); x_cost = now_cost; }
        } else {
            radius_step_rejected(&radius, &decrease_factor);
            TRACE(TR_REJECT, cand_cost, model_cost_change, rel, step_norm);
        }
        if (printSummary) printf("iter %d cost %.9e radius %.3e\n", iter, x_cost, radius);
    }
    free(r); free(J);
#undef TRACE
    if (n_iter) *n_iter = iter;
    *radius_out = radius;
    return converged && !failed;
}

static int evaluate_pnp(const void *ctx, const double x[6], double *r, double *J, double *cost) {
    return evaluate((const problem_t *)ctx, x, r, J, cost);
}

static void solve_core(float *io_state_quat, const float *cam_K, const float *pts2d, const float *pts3d,
                       const float *icov_sqrtL, int ptCnt, int maxIterCnt, float function_tolerance, int printSummary,
                       float *result_tr, int *ret, double *trace, int trace_rows, int *n_iter) {
    if (n_iter) *n_iter = 0;
    if (ptCnt < 3) { /* ceres.cpp:84-91 */
        *ret = 1;
        *result_tr = 1;
        if (printSummary) printf("skipped problem with less than 3 points\n");
        return;
    }
    problem_t P;
    P.n = ptCnt; P.u = pts2d; P.X = pts3d; P.L = icov_sqrtL;
    for (int i = 0; i < 6; ++i) P.cam[i] = cam_K[i];
    double x[6], quat[4] = {io_state_quat[0], io_state_quat[1], io_state_quat[2], io_state_quat[3]};
    quat_to_aa(quat, x); /* ceres.cpp:96 */
    for (int i = 0; i < 3; ++i) x[3 + i] = io_state_quat[4 + i];
    double radius;
    const int ok = lm_minimize(evaluate_pnp, &P, 2 * ptCnt, x, maxIterCnt, (double)function_tolerance, printSummary, &radius, trace, trace_rows,
                               n_iter);
    const int invalid = !ok;
    *ret = invalid;
    *result_tr = (float)radius;
    if (invalid) return; /* ceres.cpp:134-138: state untouched unless CONVERGENCE */
    aa_to_quat(x, quat);
    for (int i = 0; i < 4; ++i) io_state_quat[i] = (float)quat[i];
    for (int i = 0; i < 3; ++i) io_state_quat[4 + i] = (float)x[3 + i];
}

void pnp_ceres_f32(float *io_state_quat, const float *cam_K, const float *pts2d, const float *pts3d,
                   const float *icov_sqrtL, int ptCnt, int maxIterCnt, float function_tolerance, int printSummary,
                   float *result_tr, int *ret) {
    solve_core(io_state_quat, cam_K, pts2d, pts3d, icov_sqrtL, ptCnt, maxIterCnt, function_tolerance, printSummary, result_tr, ret,
               NULL, 0, NULL);
}

void pnp_ceres_f32_omp(float **init_states, float **cam_Ks, float **pts2ds, float **pts3ds, float **icov_sqrtLs,
                       int *ptCnts, int maxIterCnt, float function_tolerance, int printSummary, float *result_trs,
                       int *rets, int job_count, int num_threads) {
    if (num_threads > 1) { /* ceres.cpp:159-169 */
#ifdef _OPENMP
        omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for
        for (int i = 0; i < job_count; ++i)
            pnp_ceres_f32(init_states[i], cam_Ks[i], pts2ds[i], pts3ds[i], icov_sqrtLs[i], ptCnts[i], maxIterCnt,
                          function_tolerance, printSummary, result_trs + i, rets + i);
    } else {
        for (int i = 0; i < job_count; ++i)
            pnp_ceres_f32(init_states[i], cam_Ks[i], pts2ds[i], pts3ds[i], icov_sqrtLs[i], ptCnts[i], maxIterCnt,
                          function_tolerance, printSummary, result_trs + i, rets + i);
    }
}

/* Contiguous-batch convenience for tests/bench (same solver; avoids building pointer arrays in Python). */
void pnp_oracle_batched_f32(float *states, const float *Ks, const float *pts2d, const float *pts3d, const float *sqrtL,
                            const int *ptCnts, int nmax, int maxIterCnt, float function_tolerance, float *result_trs,
                            int *rets, int job_count, int num_threads) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for if (num_threads > 1)
    for (int i = 0; i < job_count; ++i)
        pnp_ceres_f32(states + 7 * (size_t)i, Ks + 9 * (size_t)i, pts2d + 2 * (size_t)i * nmax, pts3d + 3 * (size_t)i * nmax,
                      sqrtL + 4 * (size_t)i * nmax, ptCnts[i], maxIterCnt, function_tolerance, 0, result_trs + i, rets + i);
}

/* Same, recording the per-iteration trace: trace is (job_count, trace_rows, PNP_TRACE_COLS) doubles (zero-filled by the
 * caller), iters (job_count) receives the number of trust-region iterations of every job. */
void pnp_oracle_batched_trace_f32(float *states, const float *Ks, const float *pts2d, const float *pts3d, const float *sqrtL,
                                  const int *ptCnts, int nmax, int maxIterCnt, float function_tolerance, float *result_trs,
                                  int *rets, int job_count, int num_threads, double *trace, int trace_rows, int *iters) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for if (num_threads > 1)
    for (int i = 0; i < job_count; ++i)
        solve_core(states + 7 * (size_t)i, Ks + 9 * (size_t)i, pts2d + 2 * (size_t)i * nmax, pts3d + 3 * (size_t)i * nmax,
                   sqrtL + 4 * (size_t)i * nmax, ptCnts[i], maxIterCnt, function_tolerance, 0, result_trs + i, rets + i,
                   trace + (size_t)i * trace_rows * PNP_TRACE_COLS, trace_rows, iters + i);
}


/* ---- Published-trace check of the minimiser: Powell's function, the second example of the Ceres tutorial ("Non-linear Least
 * Squares", examples/powell.cc): f1 = x1 + 10 x2, f2 = sqrt(5) (x3 - x4), f3 = (x2 - 2 x3)^2, f4 = sqrt(10) (x1 - x4)^2 from
 * x = (3, -1, 0, 1) with DENSE_QR and default options -- the documentation prints the per-iteration log of that run (cost,
 * cost_change, |gradient|, |step|, tr_ratio, tr_radius), which tests/test_oracle_ceres_published.py compares row by row.
 * Parameters 5 and 6 are padding (see lm_minimize). */
static int evaluate_powell(const void *ctx, const double x[6], double *r, double *J, double *cost) {
    (void)ctx;
    const double s5 = sqrt(5.0), s10 = sqrt(10.0);
    r[0] = x[0] + 10.0 * x[1];
    r[1] = s5 * (x[2] - x[3]);
    r[2] = (x[1] - 2.0 * x[2]) * (x[1] - 2.0 * x[2]);
    r[3] = s10 * (x[0] - x[3]) * (x[0] - x[3]);
    *cost = 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    if (J) {
        memset(J, 0, sizeof(double) * 4 * 6);
        J[0 * 6 + 0] = 1.0; J[0 * 6 + 1] = 10.0;
        J[1 * 6 + 2] = s5; J[1 * 6 + 3] = -s5;
        J[2 * 6 + 1] = 2.0 * (x[1] - 2.0 * x[2]); J[2 * 6 + 2] = -4.0 * (x[1] - 2.0 * x[2]);
        J[3 * 6 + 0] = 2.0 * s10 * (x[0] - x[3]); J[3 * 6 + 3] = -2.0 * s10 * (x[0] - x[3]);
    }
    return isfinite(*cost);
}

/* x: 4 doubles in/out; trace: (trace_rows, PNP_TRACE_COLS); returns 1 on CONVERGENCE; *initial = {cost, max|gradient|} at the start */
int oracle_powell_trace(double *x4, int max_iter, double ftol, double *trace, int trace_rows, int *n_iter, double *radius, double *initial) {
    double x[6] = {x4[0], x4[1], x4[2], x4[3], 0.0, 0.0};
    double r[4], J[24], c;
    evaluate_powell(NULL, x, r, J, &c);
    initial[0] = c;
    initial[1] = 0;
    for (int j = 0; j < 4; ++j) {
        double gj = 0;
        for (int i = 0; i < 4; ++i) gj += J[i * 6 + j] * r[i];
        initial[1] = fmax(initial[1], fabs(gj));
    }
    const int ok = lm_minimize(evaluate_powell, NULL, 4, x, max_iter, ftol, 0, radius, trace, trace_rows, n_iter);
    for (int j = 0; j < 4; ++j) x4[j] = x[j];
    return ok;
}

/* The first example of the same tutorial ("Hello World!", examples/helloworld.cc): one residual f = 10 - x from x = 0.5.  Its
 * published log pins the Levenberg-Marquardt damping at radius 1e4 (cost 4.511598e-07 after one step) and the exit by the
 * parameter tolerance in the third iteration. */
static int evaluate_hello(const void *ctx, const double x[6], double *r, double *J, double *cost) {
    (void)ctx;
    r[0] = 10.0 - x[0];
    *cost = 0.5 * r[0] * r[0];
    if (J) { memset(J, 0, sizeof(double) * 6); J[0] = -1.0; }
    return isfinite(*cost);
}
int oracle_hello_trace(double *x1, int max_iter, double ftol, double *trace, int trace_rows, int *n_iter, double *radius) {
    double x[6] = {x1[0], 0.0, 0.0, 0.0, 0.0, 0.0};
    const int ok = lm_minimize(evaluate_hello, NULL, 1, x, max_iter, ftol, 0, radius, trace, trace_rows, n_iter);
    x1[0] = x[0];
    return ok;
}
