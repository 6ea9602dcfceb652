// Batched weighted PnP: one wavefront per pose runs the whole Levenberg-Marquardt solve in registers.
//
// Replaces lib/pnp/cxx/ceres.cpp:72-145 (pnp_ceres_f32 -> ceres::Solve, DENSE_QR, autodiff Jets) and its OpenMP
// batch driver (:147-177).  Residual model = ceres.cpp:15-65; optimiser = Ceres 2.1.0's default trust-region LM,
// restated in oracle/pnp_lm_oracle.c (see that header for the schedule and the "parity unpinned" note).
//
// MI355X mapping: lane = correspondence (grid-stride when N > 64); the pose x (angle-axis, t), the 6x6 normal
// equations and the LM state are wave-uniform values every lane carries; each evaluation reduces
// J^T J (21) | J^T r (6) | r^T r (1) with a reduce-scatter over DPP/bpermute + one LDS broadcast; the damped 6x6
// system is solved in-register (LDL^T, fp64) -- nothing but the final 7-float state goes back to HBM.
// Differences from the Ceres path that do not change the iterates beyond rounding: normal equations instead of QR
// (Jacobi-scaled, fp64, cond ~1e4..1e6), and the accepted point's Jacobian is taken from the candidate evaluation
// (same x) instead of being re-evaluated.
#include <cfloat>

#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

struct PnpPoint {
    double X[3];
    double u, v;     // measurement minus principal point (ceres.cpp:23-24)
    double a, b, c;  // L00, L10, L11 (ceres.cpp:25-27)
};

struct RotJac {
    double R[9];
    double dR[3][9];  // dR/d(aa_k)
};

// AngleAxisRotatePoint (ceres/rotation.h) as a matrix, both branches, with the exact derivative of each branch
__device__ __forceinline__ void rot_and_jac(const double aa[3], RotJac& o) {
    const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (th2 > DBL_EPSILON) {
        const double th = sqrt(th2);
        double s, c;
        sincos(th, &s, &c);
        const double ith = 1.0 / th;
        const double w[3] = {aa[0] * ith, aa[1] * ith, aa[2] * ith};
        const double oc = 1.0 - c;
        o.R[0] = c + oc * w[0] * w[0];        o.R[1] = oc * w[0] * w[1] - s * w[2]; o.R[2] = oc * w[0] * w[2] + s * w[1];
        o.R[3] = oc * w[1] * w[0] + s * w[2]; o.R[4] = c + oc * w[1] * w[1];        o.R[5] = oc * w[1] * w[2] - s * w[0];
        o.R[6] = oc * w[2] * w[0] - s * w[1]; o.R[7] = oc * w[2] * w[1] + s * w[0]; o.R[8] = c + oc * w[2] * w[2];
        const double ith2 = ith * ith;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // dR/dw_k = ((w_k [w]x + [w x (I-R) e_k]x) / |w|^2) R     (Gallego & Yezzi 2015, eq. 9)
            const double m[3] = {(k == 0) - o.R[k], (k == 1) - o.R[3 + k], (k == 2) - o.R[6 + k]};
            const double vx = aa[1] * m[2] - aa[2] * m[1], vy = aa[2] * m[0] - aa[0] * m[2], vz = aa[0] * m[1] - aa[1] * m[0];
            const double ax = (aa[k] * aa[0] + vx) * ith2, ay = (aa[k] * aa[1] + vy) * ith2, az = (aa[k] * aa[2] + vz) * ith2;
            // A = [a]x ; dR = A R
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                o.dR[k][j] = -az * o.R[3 + j] + ay * o.R[6 + j];
                o.dR[k][3 + j] = az * o.R[j] - ax * o.R[6 + j];
                o.dR[k][6 + j] = -ay * o.R[j] + ax * o.R[3 + j];
            }
        }
    } else {  // pt + aa x pt
        o.R[0] = 1; o.R[1] = -aa[2]; o.R[2] = aa[1];
        o.R[3] = aa[2]; o.R[4] = 1; o.R[5] = -aa[0];
        o.R[6] = -aa[1]; o.R[7] = aa[0]; o.R[8] = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 9; ++j) o.dR[k][j] = 0;
        o.dR[0][5] = -1; o.dR[0][7] = 1;
        o.dR[1][2] = 1; o.dR[1][6] = -1;
        o.dR[2][1] = -1; o.dR[2][3] = 1;
    }
}

// adds one correspondence's contribution to acc = [J^T J upper (21) | J^T r (6) | r^T r | pad]
__device__ __forceinline__ void accumulate_point(const PnpPoint& pt, const RotJac& rj, const double t[3],
                                                 const double k[6], double (&acc)[32]) {
    double q[3], D[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        q[d] = rj.R[3 * d] * pt.X[0] + rj.R[3 * d + 1] * pt.X[1] + rj.R[3 * d + 2] * pt.X[2] + t[d];
#pragma unroll
        for (int m = 0; m < 3; ++m)
            D[d][m] = rj.dR[m][3 * d] * pt.X[0] + rj.dR[m][3 * d + 1] * pt.X[1] + rj.dR[m][3 * d + 2] * pt.X[2];
    }
    const double iz = 1.0 / q[2];
    const double nu = q[0] * k[0] + q[1] * k[1], nv = q[0] * k[3] + q[1] * k[4];
    const double up = nu * iz, vp = nv * iz;
    const double du = up - pt.u, dv = vp - pt.v;
    const double r0 = du * pt.a + dv * pt.b, r1 = dv * pt.c;
    const double dup[3] = {k[0] * iz, k[1] * iz, -up * iz};
    const double dvp[3] = {k[3] * iz, k[4] * iz, -vp * iz};
    double J0[6], J1[6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        J0[3 + m] = pt.a * dup[m] + pt.b * dvp[m];
        J1[3 + m] = pt.c * dvp[m];
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        J0[m] = J0[3] * D[0][m] + J0[4] * D[1][m] + J0[5] * D[2][m];
        J1[m] = J1[3] * D[0][m] + J1[4] * D[1][m] + J1[5] * D[2][m];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) acc[tri6(i, j)] += J0[i] * J0[j] + J1[i] * J1[j];
        acc[21 + i] += J0[i] * r0 + J1[i] * r1;
    }
    acc[27] += r0 * r0 + r1 * r1;
}

// solve (A + diag(dg)) y = rhs for symmetric A (packed upper 21) by LDL^T; false if a pivot is not positive/finite
__device__ __forceinline__ bool ldlt_solve6(const double (&A)[21], const double (&dg)[6], const double (&rhs)[6], double (&y)[6]) {
    double L[6][6], d[6], id[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = A[tri6(j, j)] + dg[j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
        d[j] = dj;
        ok = ok && (dj > 0) && (dj < DBL_MAX);
        id[j] = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = A[tri6(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * id[j];
        }
    }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = rhs[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= L[i][k] * z[k];
        z[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = z[i] * id[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= L[k][i] * y[k];
        y[i] = v;
    }
    return ok;
}

__device__ __forceinline__ PnpPoint load_point(const PnpParams& p, size_t base, int n, const double cam[6]) {
    PnpPoint o;
    const float* X = p.pts3d + (base + n) * 3;
    const float2 u = *reinterpret_cast<const float2*>(p.pts2d + (base + n) * 2);
    o.X[0] = X[0]; o.X[1] = X[1]; o.X[2] = X[2];
    o.u = (double)u.x - cam[2];
    o.v = (double)u.y - cam[5];
    if (p.sqrtL) {
        const float4 L = *reinterpret_cast<const float4*>(p.sqrtL + (base + n) * 4);
        o.a = L.x; o.b = L.z; o.c = L.w;
    } else {
        const float2 L = *reinterpret_cast<const float2*>(p.sqrt_diag + (base + n) * 2);
        o.a = L.x; o.b = 0; o.c = L.y;
    }
    return o;
}

template <bool REG>
__global__ __launch_bounds__(64) void lc_pnp_lm_kernel(const PnpParams p) {
    __shared__ double bc[32];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = p.counts ? p.counts[b] : p.Nmax;
    if (n < 3) {  // ceres.cpp:84-91
        if (lane == 0) {
            p.rets[b] = 1;
            p.result_tr[b] = 1.f;
            if (p.iters) p.iters[b] = 0;
        }
        return;
    }
    const size_t base = (size_t)b * p.Nmax;
    double cam[6];
    {
        const float* Kp = p.K + 9 * (size_t)b;
#pragma unroll
        for (int i = 0; i < 6; ++i) cam[i] = Kp[i];  // only the first 6 floats are read (ceres.cpp:99-101)
    }
    double x[6];
    {
        // QuaternionToAngleAxis (ceres.cpp:96)
        const float* st = p.states + 7 * (size_t)b;
        const double q0 = st[0], q1 = st[1], q2 = st[2], q3 = st[3];
        const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
        double kk = 2.0;
        if (s2 > 0.0) {
            const double s = sqrt(s2);
            const double two_theta = 2.0 * ((q0 < 0.0) ? atan2(-s, -q0) : atan2(s, q0));
            kk = two_theta / s;
        }
        x[0] = q1 * kk; x[1] = q2 * kk; x[2] = q3 * kk;
        x[3] = st[4]; x[4] = st[5]; x[5] = st[6];
    }
    PnpPoint rp;
    const bool active = lane < n;
    if constexpr (REG) {
        if (active) rp = load_point(p, base, lane, cam);
    }

    // full evaluation at xe: H (21), g (6), cost; returns false when anything is non-finite
    auto evaluate = [&](const double (&xe)[6], double (&H)[21], double (&g)[6], double& cost) -> bool {
        RotJac rj;
        rot_and_jac(xe, rj);
        const double t[3] = {xe[3], xe[4], xe[5]};
        double acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = 0;
        if constexpr (REG) {
            if (active) accumulate_point(rp, rj, t, cam, acc);
        } else {
            for (int i = lane; i < n; i += kWave) accumulate_point(load_point(p, base, i, cam), rj, t, cam, acc);
        }
        wave_reduce_scatter16<32>(acc, lane);
        __syncthreads();  // previous broadcast fully consumed
        if ((lane & 3) == 0) {
            const int bs = scatter16_base(lane, 2);
            bc[bs] = acc[0];
            bc[bs + 1] = acc[1];
        }
        __syncthreads();
        bool fin = true;
#pragma unroll
        for (int i = 0; i < 21; ++i) { H[i] = bc[i]; fin = fin && (fabs(H[i]) <= DBL_MAX); }
#pragma unroll
        for (int i = 0; i < 6; ++i) { g[i] = bc[21 + i]; fin = fin && (fabs(g[i]) <= DBL_MAX); }
        const double ss = bc[27];
        cost = 0.5 * ss;
        return fin && (fabs(ss) <= DBL_MAX);
    };

    const double ftol = p.ftol, ptol = 1e-8, gtol = 1e-10;
    double H[21], g[6], cost;
    bool failed = !evaluate(x, H, g, cost);
    double scale[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) scale[j] = 1.0 / (1.0 + sqrt(H[tri6(j, j)]));  // Jacobi scaling, fixed at iteration 0
    auto max_abs6 = [](const double (&v)[6]) {
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m = fmax(m, fabs(v[j]));
        return m;
    };
    auto norm6 = [](const double (&v)[6]) {
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m += v[j] * v[j];
        return sqrt(m);
    };
    double gmax = max_abs6(g), xnorm = norm6(x);
    double radius = 1e4, dfac = 2.0;
    int iter = 0, n_invalid = 0;
    bool converged = false;

    while (!failed && !converged) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (iter >= p.max_iter) break;
        if (gmax <= gtol || radius <= 1e-32) { converged = true; break; }
        ++iter;
        // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system
        double As[21], dg[6], rhs[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) As[tri6(i, j)] = H[tri6(i, j)] * scale[i] * scale[j];
            rhs[i] = g[i] * scale[i];
            dg[i] = fmin(fmax(As[tri6(i, i)], 1e-6), 1e32) / radius;
        }
        bool step_ok = ldlt_solve6(As, dg, rhs, y);
        double mcc = 0;  // model_cost_change = y.(Js^T r) - y^T Js^T Js y / 2   (step = -y)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double row = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) row += As[i <= j ? tri6(i, j) : tri6(j, i)] * y[j];
            mcc += y[i] * (rhs[i] - 0.5 * row);
            step_ok = step_ok && (fabs(y[i]) <= DBL_MAX);
        }
        step_ok = step_ok && (mcc > 0.0);
        if (!step_ok) {  // HandleInvalidStep
            if (++n_invalid >= 5) { failed = true; break; }
            radius /= dfac; dfac *= 2.0;
            continue;
        }
        n_invalid = 0;
        double xc[6], delta[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { delta[j] = -y[j] * scale[j]; xc[j] = x[j] + delta[j]; }
        const double step_norm = norm6(delta);
        double Hc[21], gc[6], cost_c;
        const bool cand_ok = evaluate(xc, Hc, gc, cost_c);
        if (!cand_ok) cost_c = DBL_MAX;
        if (step_norm <= ptol * (xnorm + ptol)) { converged = true; break; }  // ParameterToleranceReached
        const double cost_change = cost - cost_c;
        if (fabs(cost_change) <= ftol * cost) { converged = true; break; }    // FunctionToleranceReached
        const double rel = cost_change / mcc;
        if (rel > 1e-3) {  // HandleSuccessfulStep
#pragma unroll
            for (int j = 0; j < 6; ++j) { x[j] = xc[j]; g[j] = gc[j]; }
#pragma unroll
            for (int j = 0; j < 21; ++j) H[j] = Hc[j];
            cost = cost_c;
            xnorm = norm6(x);
            gmax = max_abs6(g);
            const double tq = 2.0 * rel - 1.0;
            radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - tq * tq * tq));
            dfac = 2.0;
        } else {
            radius /= dfac; dfac *= 2.0;
        }
    }
    const bool invalid = failed || !converged;
    if (lane == 0) {
        p.rets[b] = invalid ? 1 : 0;
        p.result_tr[b] = (float)radius;
        if (p.iters) p.iters[b] = iter;
        if (!invalid) {  // ceres.cpp:131-144: AngleAxisToQuaternion, write back in place
            float* st = p.states + 7 * (size_t)b;
            const double t2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
            double q0 = 1.0, kk = 0.5;
            if (t2 > 0.0) {
                const double th = sqrt(t2), h = 0.5 * th;
                double sh, ch;
                sincos(h, &sh, &ch);
                q0 = ch;
                kk = sh / th;
            }
            st[0] = (float)q0; st[1] = (float)(x[0] * kk); st[2] = (float)(x[1] * kk); st[3] = (float)(x[2] * kk);
            st[4] = (float)x[3]; st[5] = (float)x[4]; st[6] = (float)x[5];
        }
    }
}

}  // namespace

int launch_pnp_lm(const PnpParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.Nmax <= 64)
        hipLaunchKernelGGL(lc_pnp_lm_kernel<true>, dim3(p.B), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL(lc_pnp_lm_kernel<false>, dim3(p.B), dim3(64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
