// Device body of the batched weighted-PnP solve: one wavefront runs one pose's whole Levenberg-Marquardt solve.
// Included by lc_pnp.hip (stand-alone kernel) and lc_fused.hip (loss + PnP in one launch).
//
// Replaces lib/pnp/cxx/ceres.cpp:72-145 (pnp_ceres_f32 -> ceres::Solve, DENSE_QR, autodiff Jets) and its OpenMP
// batch driver (:147-177).  Residual model = ceres.cpp:15-65; optimiser = Ceres 2.1.0's default trust-region LM,
// restated in oracle/pnp_lm_oracle.c (see that header for the schedule and the "parity unpinned" note).
//
// MI355X mapping: lane = correspondence (wave-stride when N > 64); the pose x = (angle-axis, t), the 6x6 normal
// equations and the LM state are wave-uniform values every lane carries.  One evaluation per LM iteration:
//   residuals in fp64 from R(x) X + t;  Jacobian rows from the closed form  J_rot = Jr(w) (R X x J_t)
//   (d(R(w)X)/dw = -R [X]x Jr(w) = -[R X]x Jr(w)^T, Jr = right Jacobian of SO(3)) -- no per-point 3x3 derivative, no autodiff;
//   J^T J (21) | J^T r (6) | r^T r (1) reduced with permlane-swap/DPP reduce-scatter + ONE LDS broadcast;
//   damped, Jacobi-scaled 6x6 system solved in registers (LDL^T, fp64).
// Only the final 7-float state, the trust radius and the flag go back to HBM.
// Differences from the Ceres path that do not change the iterates beyond rounding: normal equations instead of QR
// (Jacobi-scaled, fp64, cond ~1e4..1e6); the accepted point's Jacobian comes from the candidate evaluation (same x)
// instead of a re-evaluation; model_cost_change from the normal-equation identity y.(g/2) + sum(d y^2)/2.
#pragma once
#include <cfloat>

#include "lc_common.h"
#include "lc_kernels.h"

// Diagnostic build only (-DLC_STAMPS, scripts/diag_stamps.py): per-phase s_memtime sums, written to p.result_tr's
// neighbour buffer p.iters (reinterpreted) which nothing else reads in that build.  Never defined in the shipped library.
#ifdef LC_STAMPS
#define LC_PSTAMP_DECL unsigned long long pst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0_ = 0, pt1_ = 0
#define LC_PSTAMP_BEGIN()                                                                 \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#define LC_PSTAMP(i)                                                                      \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt1_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                \
        pst_[i] += pt1_ - pt0_;                                                           \
        pt0_ = pt1_;                                                                      \
    } while (0)
#else
#define LC_PSTAMP_DECL
#define LC_PSTAMP_BEGIN() do {} while (0)
#define LC_PSTAMP(i) do {} while (0)
#endif

#ifndef LC_WIDE_REG_SUM_REGS
#define LC_WIDE_REG_SUM_REGS 1  // A/B switch: the four-wave register block sum also on the one-correspondence-per-thread path (N <= 256: 12.4 -> 11.6 us at 64 x 256; 0 = sums streamed through LDS)
#endif
#ifndef LC_PNP_STREAM_SUM
#define LC_PNP_STREAM_SUM 1  // A/B switch of the streamed block sum (scripts/ubench/pnp_ab.py); 1 in the shipped library
#endif

namespace lc {
namespace pnp {

template <int NW>
constexpr int kPnpLdsDoubles = sum_bcast_lds_doubles<NW>(28);  // block_sum_bcast_lds<28, NW>

struct Point {
    double X[3];
    double u, v;     // measurement minus principal point (ceres.cpp:23-24)
    double a, b, c;  // L00, L10, L11 (ceres.cpp:25-27)
};

struct Rot {
    double R[9];   // AngleAxisRotatePoint as a matrix (both branches of ceres/rotation.h)
    double Jr[9];  // right Jacobian of SO(3) (identity in the small-angle branch, whose derivative is -R[pt]x with
                   // R = I + [aa]x, |aa| < 1.5e-8: within 1e-8 of the -[pt]x the autodiff of that branch yields)
};

__device__ __forceinline__ void make_rot(const double aa[3], Rot& o) {
    // branch-free on purpose: a two-branch version keeps `o` in scratch memory (104 B/lane of HBM traffic per evaluation)
    const double x = aa[0], y = aa[1], z = aa[2];
    const double th2 = x * x + y * y + z * z;
    const bool small = !(th2 > DBL_EPSILON);  // ceres/rotation.h AngleAxisRotatePoint: pt + aa x pt, derivative -[pt]x
    double th, ith;
    fast_sqrt_rsqrt(th2, th, ith);
    double s, c;
    sincos_small(th, s, c);
    const double ith2 = ith * ith;
    const double A = small ? 1.0 : s * ith;
    const double B = small ? 0.0 : (1.0 - c) * ith2;
    const double C = small ? 0.0 : (th - s) * ith2 * ith;
    const double cd = small ? 1.0 : c;
    // R = c I + A [w]x + B w w^T
    o.R[0] = cd + B * x * x;    o.R[1] = B * x * y - A * z; o.R[2] = B * x * z + A * y;
    o.R[3] = B * x * y + A * z; o.R[4] = cd + B * y * y;    o.R[5] = B * y * z - A * x;
    o.R[6] = B * x * z - A * y; o.R[7] = B * y * z + A * x; o.R[8] = cd + B * z * z;
    // Jr = (1 - C th2) I - B [w]x + C w w^T   (identity in the small-angle branch)
    const double d = 1.0 - C * th2;
    o.Jr[0] = d + C * x * x;     o.Jr[1] = C * x * y + B * z; o.Jr[2] = C * x * z - B * y;
    o.Jr[3] = C * x * y - B * z; o.Jr[4] = d + C * y * y;     o.Jr[5] = C * y * z + B * x;
    o.Jr[6] = C * x * z + B * y; o.Jr[7] = C * y * z - B * x; o.Jr[8] = d + C * z * z;
}

// adds one correspondence's contribution to acc = [J^T J upper (21) | J^T r (6) | r^T r | pad]
// (columns of J pre-multiplied by the Jacobi scaling sc: acc holds Js^T Js and Js^T r directly)
// FIRST: acc is written (not added to) -- saves zero-filling 32 accumulators and one add per entry when a lane owns one point.
// STREAM: (FIRST only) every finished entry goes straight to its LDS slot of the block sum instead of into acc[]
template <bool FIRST, bool STREAM = false, int NW = 1>
__device__ __forceinline__ void accumulate_point(const Point& pt, const Rot& rt, const double t[3], const double k[6],
                                                 const double (&sc)[6], double (&acc)[28], double* lds = nullptr, int pos = 0) {
    double rx[3], q[3];  // R X and R X + t
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rx[d] = rt.R[3 * d] * pt.X[0] + rt.R[3 * d + 1] * pt.X[1] + rt.R[3 * d + 2] * pt.X[2];
        q[d] = rx[d] + t[d];
    }
    const double iz = fast_rcp(q[2]);
    const double up = (q[0] * k[0] + q[1] * k[1]) * iz, vp = (q[0] * k[3] + q[1] * k[4]) * iz;
    const double du = up - pt.u, dv = vp - pt.v;
    const double r[2] = {du * pt.a + dv * pt.b, dv * pt.c};
    const double dup[3] = {k[0] * iz, k[1] * iz, -up * iz};
    const double dvp[3] = {k[3] * iz, k[4] * iz, -vp * iz};
    double J[2][6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        J[0][3 + m] = pt.a * dup[m] + pt.b * dvp[m];
        J[1][3 + m] = pt.c * dvp[m];
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        // d(R(w) X)/dw = -R [X]x Jr(w) = -[R X]x Jl(w) with the left Jacobian Jl = R Jr = Jr^T: the row a = J_t of the residual gives
        // J_rot = Jl^T (R X x a) = Jr (R X x a) -- R X is at hand from the projection, so no R^T a is formed (9 fma fewer per row)
        double cr[3];
        cr[0] = rx[1] * J[rr][5] - rx[2] * J[rr][4];
        cr[1] = rx[2] * J[rr][3] - rx[0] * J[rr][5];
        cr[2] = rx[0] * J[rr][4] - rx[1] * J[rr][3];
#pragma unroll
        for (int m = 0; m < 3; ++m) J[rr][m] = (rt.Jr[3 * m] * cr[0] + rt.Jr[3 * m + 1] * cr[1] + rt.Jr[3 * m + 2] * cr[2]) * sc[m];  // Jr (R X x a)
#pragma unroll
        for (int m = 0; m < 3; ++m) J[rr][3 + m] *= sc[3 + m];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
            const double v = __builtin_fma(J[0][i], J[0][j], J[1][i] * J[1][j]);
            if constexpr (STREAM) block_sum_put<NW>(lds, pos, tri6(i, j), v);
            else acc[tri6(i, j)] = FIRST ? v : acc[tri6(i, j)] + v;
        }
        const double gi = __builtin_fma(J[0][i], r[0], J[1][i] * r[1]);
        if constexpr (STREAM) block_sum_put<NW>(lds, pos, 21 + i, gi);
        else acc[21 + i] = FIRST ? gi : acc[21 + i] + gi;
    }
    const double ss = __builtin_fma(r[0], r[0], r[1] * r[1]);
    if constexpr (STREAM) block_sum_put<NW>(lds, pos, 27, ss);
    else acc[27] = FIRST ? ss : acc[27] + ss;
}

// solve (A + diag(dg)) y = rhs for symmetric A (packed upper 21) by LDL^T; false if a pivot is not positive/finite
__device__ __forceinline__ bool ldlt_solve6(const double (&A)[21], const double (&dg)[6], const double (&rhs)[6], double (&y)[6]) {
    double L[6][6], Ld[6][6], id[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = A[tri6(j, j)] + dg[j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * Ld[j][k];
        ok = ok && (dj > 0) && (dj < DBL_MAX);
        id[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = A[tri6(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * Ld[j][k];
            Ld[i][j] = v;            // L[i][j] * d[j]
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

// one correspondence as it sits in HBM (fp32): the loads are issued first thing in the kernel, the conversion to the fp64 working
// form waits until the pose-independent set-up (quaternion -> angle-axis, ~700 cycles) has run in their shadow
struct RawPoint {
    float X[3];
    float2 u;
    float a, b, c;
};
// torch.nan_to_num with its defaults: NaN -> 0, +-inf -> +-FLT_MAX (cer_solver.py:29-31 applies it to every input when asked to)
__device__ __forceinline__ float nan_to_num(float f) { return f != f ? 0.f : fminf(fmaxf(f, -FLT_MAX), FLT_MAX); }

// OPTS: the instantiation that honours PnpParams::options / weight_mask (input filtering and weight forms of the callers, done at
// the load instead of by separate element-wise launches); the plain instantiation does not even test the fields
template <bool OPTS = false>
__device__ __forceinline__ RawPoint load_raw_point(const PnpParams& p, size_t base, int n) {
    RawPoint o;
    const float* X = p.pts3d + (base + n) * 3;
    o.u = *reinterpret_cast<const float2*>(p.pts2d + (base + n) * 2);
    o.X[0] = X[0]; o.X[1] = X[1]; o.X[2] = X[2];
    if (OPTS && p.weight_mask) {  // unit information on the flagged correspondences, none on the others
        const float m = p.weight_mask[base + n] ? 1.f : 0.f;
        o.a = m; o.b = 0.f; o.c = m;
    } else if (p.sqrtL) {
        const float4 L = *reinterpret_cast<const float4*>(p.sqrtL + (base + n) * 4);
        o.a = L.x; o.b = L.z; o.c = L.w;
    } else {
        const float2 L = *reinterpret_cast<const float2*>(p.sqrt_diag + (base + n) * 2);
        o.a = L.x; o.b = 0.f; o.c = L.y;
    }
    if constexpr (OPTS) {
        if (p.options & kPnpWeightsAreStd) { o.a = 1.f / (o.a * o.a); o.c = 1.f / (o.c * o.c); }  // the sparse head's predicted deviations -> inverse variances (test.py:52)
        if (p.options & kPnpNanToNum) {
            o.u.x = nan_to_num(o.u.x); o.u.y = nan_to_num(o.u.y);
            o.X[0] = nan_to_num(o.X[0]); o.X[1] = nan_to_num(o.X[1]); o.X[2] = nan_to_num(o.X[2]);
            o.a = nan_to_num(o.a); o.b = nan_to_num(o.b); o.c = nan_to_num(o.c);
        }
        if (p.options & kPnpWeightsAreIcov) { o.a = sqrtf(o.a); o.c = sqrtf(o.c); }  // diagonal inverse covariance -> its factor (cer_solver.py:33-36)
    }
    return o;
}
__device__ __forceinline__ Point to_point(const RawPoint& r, const double cam[6]) {
    Point o;
    o.X[0] = r.X[0]; o.X[1] = r.X[1]; o.X[2] = r.X[2];
    o.u = (double)r.u.x - cam[2];
    o.v = (double)r.u.y - cam[5];
    o.a = r.a; o.b = r.b; o.c = r.c;
    return o;
}
template <bool OPTS = false>
__device__ __forceinline__ Point load_point(const PnpParams& p, size_t base, int n, const double cam[6]) {
    return to_point(load_raw_point<OPTS>(p, base, n), cam);
}

__device__ __forceinline__ double max_abs6(const double (&v)[6]) {
    double m = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) m = fmax(m, fabs(v[j]));
    return m;
}
__device__ __forceinline__ double norm6(const double (&v)[6]) {
    double m = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) m += v[j] * v[j];
    return fast_sqrt(m);
}

// One pose per workgroup of NW wavefronts (NW = 1: N <= 64, the metric's shape; NW = 4: larger N, e.g. the dense heads'
// N = 1024..1849, where four waves cut the per-evaluation point loop by four; the wave-uniform LM algebra is simply
// replicated in every wave).  bc: kPnpLdsDoubles<NW> doubles of LDS.
// REG: Nmax <= 64*NW, each thread keeps its correspondence in registers across the whole solve.
// TRACE: diagnostic instantiation (lc_pnp_lm_trace_f32) that also writes one row per trust-region iteration to p.trace, in the
// column layout of oracle/pnp_lm_oracle.c's PNP_TRACE_COLS; the shipped kernels are instantiated with TRACE = false.
// PPT (with !REG): a thread keeps its first PPT correspondences (lane, lane + 64 NW, ...) in registers as they sit in HBM (8 floats
// each) across the whole solve; the block-stride loop otherwise re-reads them from L2 in every evaluation, ~1 us of exposed latency
// each time.  TAIL: correspondences behind the cached prefix (n > 64 NW PPT) are read from memory; instantiations whose rows always fit
// the prefix leave it out (the never-taken loop cost the test-time chain 1 us).  Same accumulation order, same results.
// store / handoff / start_override (the chained launch, lc_pnp.hip): store = false keeps the job's outputs in registers (a workgroup that
// repeats a solve another workgroup owns must not write the owner's rows a second time); handoff (7 floats of LDS) receives the state the
// solve would leave in p.states[b]; start_override replaces the start pose (7 floats, e.g. a previous solve's handoff).
// SPLIT = 1 (NW = 4, PPT > 0): workgroup sx->part of the sx->G that share pose b -- it owns the correspondences part * 64 NW + lane + k * 64 NW G,
// the block sum runs across the G workgroups (lc_common.h: block_sum_split), everything else is replicated; part 0 stores -- status 2 when
// its wait for another part ran out.  SPLIT = 2 (NW = 4, PPT = 0): ONE workgroup plays the sx->G parts in turn (the rescue launch of a pose
// of status 2: block_sum_parts_serial) -- per thread the same correspondences in the same order, every sum in the same order: the same bits.
template <bool REG, int NW = 1, bool TRACE = false, bool OPTS = false, int PPT = 0, bool TAIL = false, int SPLIT = 0>
__device__ __forceinline__ void solve_pose(const PnpParams& p, int b, int lane, double* bc, bool store = true, float* handoff = nullptr,
                                           const float* start_override = nullptr, [[maybe_unused]] SplitSum* sx = nullptr) {
    constexpr int kThreads = kWave * NW;  // `lane` is the thread index within the workgroup
    static_assert(SPLIT != 1 || (NW == 4 && !REG && PPT > 0 && !TRACE), "the split form is the four-wave cached-prefix solve");
    static_assert(SPLIT != 2 || (NW == 4 && !REG && PPT == 0 && !TRACE), "the rescue form walks memory, part by part");
    const int slot = SPLIT == 1 ? lane + kThreads * sx->part : lane;  // first correspondence of this thread, and the distance to its next
    const int stride = SPLIT == 1 ? kThreads * sx->G : kThreads;
#ifdef LC_TRACE_CLOCK
    const unsigned long long t_start_ = __builtin_amdgcn_s_memtime();
#endif
    LC_PSTAMP_DECL;
    LC_PSTAMP_BEGIN();
    const int n = p.counts ? p.counts[b] : p.Nmax;
    // pose_mod (OPTS launches): K and start hold pose_mod rows shared by the poses b, b + pose_mod, ... (several solves of the same
    // objects -- different correspondence selections -- batched into one launch)
    const int bk = (OPTS && p.pose_mod > 0) ? b % p.pose_mod : b;
    const float* st_in = start_override ? start_override : (p.start ? p.start + 7 * (size_t)bk : p.states + 7 * (size_t)b);
    const bool filter = OPTS && (p.options & kPnpNanToNum);
    auto fin = [&](float f) { return filter ? nan_to_num(f) : f; };
    if (n < 3) {  // ceres.cpp:84-91
        if (lane == 0 && store) {
            p.rets[b] = 1;
            p.result_tr[b] = 1.f;
            if (p.iters) p.iters[b] = 0;
        }
        if (lane < 7) {
            const float v = fin(st_in[lane]);
            if (store && (p.start || start_override || filter)) p.states[7 * (size_t)b + lane] = v;
            if (handoff) handoff[lane] = v;
        }
        return;
    }
    const size_t base = (size_t)b * p.Nmax;
    const bool active = lane < n;
    RawPoint raw;
    if constexpr (REG) raw = load_raw_point<OPTS>(p, base, active ? lane : 0);  // n >= 3 here: correspondence 0 exists
    [[maybe_unused]] RawPoint rawc[PPT > 0 ? PPT : 1];
    if constexpr (!REG && PPT > 0) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (slot + k * stride < n) rawc[k] = load_raw_point<OPTS>(p, base, slot + k * stride);
    }
    double cam[6];
    {
        const float* Kp = p.K + 9 * (size_t)bk;
#pragma unroll
        for (int i = 0; i < 6; ++i) cam[i] = fin(Kp[i]);  // only the first 6 floats are read (ceres.cpp:99-101)
    }
    double x[6];
    {
        // QuaternionToAngleAxis (ceres.cpp:96)
        // q0 is only used inside `if (s2 > 0)`, and the compiler sinks its LOAD in there: a second memory round trip after the one that
        // brought q1..q3 of the same cache line (seen in the ISA of the pose unit: global_load of start[0] behind the first s_waitcnt).  The
        // empty asm is a use of the seven loaded values in the straight-line code: their loads are issued with the others (with q0 alone
        // pinned, the compiler moved q1..q3 behind the wait instead).
        float stf[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) stf[i] = st_in[i];
        asm volatile("" : "+v"(stf[0]), "+v"(stf[1]), "+v"(stf[2]), "+v"(stf[3]), "+v"(stf[4]), "+v"(stf[5]), "+v"(stf[6]));
        const double q0 = fin(stf[0]), q1 = fin(stf[1]), q2 = fin(stf[2]), q3 = fin(stf[3]);
        const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
        double kk = 2.0;
        if (s2 > 0.0) {
            const double s = fast_sqrt(s2);
            // ceres/rotation.h: 2 atan2(s, c) with c >= 0, or 2 atan2(-s, -c) with c < 0: both are +-2 atan(s / |c|), s > 0
            const double half = atan_ratio_pos(s, fabs(q0));
            const double two_theta = 2.0 * ((q0 < 0.0) ? -half : half);
            kk = two_theta / s;
        }
        x[0] = q1 * kk; x[1] = q2 * kk; x[2] = q3 * kk;
        x[3] = fin(stf[4]); x[4] = fin(stf[5]); x[5] = fin(stf[6]);
    }
    LC_PSTAMP(0);
    // Lanes without a correspondence (lane >= n) carry a copy of correspondence 0 with a ZERO information factor: their residuals
    // and Jacobian rows are exact zeros, so they add nothing to the sums and need no masking in the evaluation (56 selects per
    // evaluation before); their projection is finite exactly when correspondence 0's is, and a non-finite correspondence fails the
    // job either way (ceres.cpp:126-138 -> invalid).
    Point rp;
    if constexpr (REG) {
        rp = to_point(raw, cam);
        if (!active) { rp.a = 0.0; rp.b = 0.0; rp.c = 0.0; }
    }

    [[maybe_unused]] int sum_phase = 0;  // block_sum_waves4: which of its two rows of totals the next sum writes
    // full evaluation at xe with column scaling sc: H = Js^T Js (21), g = Js^T r (6), cost; false when anything is non-finite
    auto evaluate = [&](const double (&xe)[6], const double (&sc)[6], double (&H)[21], double (&g)[6], double& cost) -> bool {
        LC_PSTAMP(1);
        Rot rt;
        make_rot(xe, rt);
        LC_PSTAMP(2);
        const double t[3] = {xe[3], xe[4], xe[5]};
        double acc[28];
        if constexpr (REG && LC_PNP_STREAM_SUM && !(NW == 4 && LC_WIDE_REG_SUM_REGS)) {
            // lanes beyond n hold zero-weight copies (exact zeros); the 28 partial sums go to LDS as they are produced
            const int pos = block_sum_open<NW>(lane);
            accumulate_point<true, true, NW>(rp, rt, t, cam, sc, acc, bc, pos);
            LC_PSTAMP(3);
            block_sum_close<28, NW>(acc, bc, lane);
        } else if constexpr (REG) {
            accumulate_point<true>(rp, rt, t, cam, sc, acc);
            LC_PSTAMP(3);
            if constexpr (NW == 4 && LC_WIDE_SUM_REGS) block_sum_waves4<28>(acc, bc, lane, sum_phase);
            else block_sum_bcast_lds<28, NW>(acc, bc, lane);
        } else {
#pragma unroll
            for (int i = 0; i < 28; ++i) acc[i] = 0;
            if constexpr (PPT > 0) {
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    if (slot + k * stride < n) accumulate_point<false>(to_point(rawc[k], cam), rt, t, cam, sc, acc);
                // rows wider than the cached prefix (Nmax > 64 NW PPT): the correspondences behind it from memory, in the same per-thread
                // order as the plain loop below -- same sums bit for bit; empty when the pose's count fits the prefix
                if constexpr (TAIL)
                    for (int i = slot + PPT * stride; i < n; i += stride) accumulate_point<false>(load_point<OPTS>(p, base, i, cam), rt, t, cam, sc, acc);
            } else if constexpr (SPLIT == 2) {
                for (int g = 0; g < sx->G; ++g) {  // part g's share and its totals; the last call adds the parts in order
                    if (g > 0) {
#pragma unroll
                        for (int i = 0; i < 28; ++i) acc[i] = 0;
                    }
                    for (int i = lane + kThreads * g; i < n; i += kThreads * sx->G) accumulate_point<false>(load_point<OPTS>(p, base, i, cam), rt, t, cam, sc, acc);
                    block_sum_parts_serial<28>(acc, bc, lane, g, sx->G);
                }
            } else {
                for (int i = lane; i < n; i += kThreads) accumulate_point<false>(load_point<OPTS>(p, base, i, cam), rt, t, cam, sc, acc);
            }
            LC_PSTAMP(3);
            if constexpr (SPLIT == 2) {}
            else if constexpr (SPLIT == 1) block_sum_split<28>(acc, bc, lane, sum_phase, *sx);
            else if constexpr (NW == 4 && LC_WIDE_SUM_REGS) block_sum_waves4<28>(acc, bc, lane, sum_phase);
            else block_sum_bcast_lds<28, NW>(acc, bc, lane);
        }
        LC_PSTAMP(4);
#pragma unroll
        for (int i = 0; i < 21; ++i) H[i] = acc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = acc[21 + i];
        const double ss = acc[27];
        cost = 0.5 * ss;
        double chk = ss;  // every term is >= 0: the sum is finite iff all residuals and Jacobian entries are
#pragma unroll
        for (int i = 0; i < 6; ++i) chk += H[tri6(i, i)];
        LC_PSTAMP(5);
        if constexpr (SPLIT == 1) {
            if (sx->timed_out) return false;  // a workgroup of this pose never arrived: this launch gives the pose up (lc_common.h: SplitSum)
        }
        return chk <= DBL_MAX;
    };

    const double ftol = p.ftol, ptol = 1e-8, gtol = 1e-10;
    double H[21], g[6], cost;  // of the Jacobi-scaled problem from here on
    double scale[6], iscale[6];
    bool failed;
    {
        const double one[6] = {1, 1, 1, 1, 1, 1};
        failed = !evaluate(x, one, H, g, cost);
#pragma unroll
        for (int j = 0; j < 6; ++j) {  // Jacobi scaling 1/(1+||J_j||), fixed at iteration 0
            iscale[j] = 1.0 + fast_sqrt(H[tri6(j, j)]);  // 4e-15 relative (lc_common.h); the libm sqrt costs ~100 cycles x 6 on the start-up path
            scale[j] = fast_rcp(iscale[j]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) H[tri6(i, j)] *= scale[i] * scale[j];
            g[i] *= scale[i];
        }
    }
    auto grad_max = [&](const double (&gs)[6]) {  // max-norm of the UNSCALED gradient J^T r
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m = fmax(m, fabs(gs[j]) * iscale[j]);
        return m;
    };
    double gmax = grad_max(g);
    auto sqnorm6 = [](const double (&v)[6]) {
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m += v[j] * v[j];
        return m;
    };
    double xn2 = sqnorm6(x);  // ||x||^2; the norm itself is only formed when the parameter-tolerance test is in reach
    double radius = 1e4, dfac = 2.0;
    int iter = 0, n_invalid = 0;
    bool converged = false;
    // kind: 0 invalid step, 1 accepted, 2 rejected, 3 parameter tolerance, 4 function tolerance
    auto trace = [&](int kind, double cost_x, double cost_cand, double mcc, double rho, double step_norm) {
        if constexpr (TRACE) {
            if (lane == 0 && p.trace && iter <= p.trace_rows) {
                double* row = p.trace + ((size_t)b * p.trace_rows + (iter - 1)) * 8;
                row[0] = kind; row[1] = cost_x; row[2] = cost_cand; row[3] = mcc; row[4] = rho; row[5] = step_norm;
                row[6] = radius; row[7] = gmax;
#ifdef LC_TRACE_CLOCK  // timing diagnostics only (scripts/ubench/pnp_iter_clock.py): shader cycles since the wave started
                row[7] = (double)(__builtin_amdgcn_s_memtime() - t_start_);
#endif
            }
        }
    };

    while (uniform(!failed && !converged)) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (iter >= p.max_iter) break;
        if (uniform(gmax <= gtol || radius <= 1e-32)) { converged = true; break; }
        ++iter;
        // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system
        double dg[6], y[6];
        const double inv_radius = fast_rcp(radius);
#pragma unroll
        for (int i = 0; i < 6; ++i) dg[i] = fmin(fmax(H[tri6(i, i)], 1e-6), 1e32) * inv_radius;
        LC_PSTAMP(1);
        bool step_ok = ldlt_solve6(H, dg, g, y);
        LC_PSTAMP(6);
        // model_cost_change = y.g - y^T H y / 2 with (H + D) y = g  =>  (y.g + sum d_i y_i^2) / 2   (step = -y)
        double mcc = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) mcc += y[i] * (g[i] + dg[i] * y[i]);
        mcc *= 0.5;
        step_ok = step_ok && (mcc > 0.0) && (mcc <= DBL_MAX);  // a non-finite y makes mcc non-finite
        if (uniform(!step_ok)) {  // HandleInvalidStep
            if (++n_invalid >= 5) { failed = true; trace(0, cost, 0.0, mcc, 0.0, 0.0); break; }
            radius /= dfac; dfac *= 2.0;
            trace(0, cost, 0.0, mcc, 0.0, 0.0);
            continue;
        }
        n_invalid = 0;
        double xc[6], delta[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { delta[j] = -y[j] * scale[j]; xc[j] = x[j] + delta[j]; }
        // ParameterToleranceReached is  ||delta|| <= ptol (||x|| + ptol).  Since (a + b)^2 <= 2 (a^2 + b^2), a step with
        // ||delta||^2 > 2 ptol^2 (||x||^2 + ptol^2) cannot pass: two square roots per iteration are only taken next to convergence
        // (and by the diagnostic twin, which records the step norm).
        const double sn2 = sqnorm6(delta);
        const bool ptol_in_reach = TRACE || uniform(!(sn2 > 2.0 * ptol * ptol * (xn2 + ptol * ptol)));
        double step_norm = 0.0;
        bool ptol_hit = false;
        if (ptol_in_reach) {  // scalar branch
            step_norm = fast_sqrt(sn2);
            ptol_hit = uniform(step_norm <= ptol * (fast_sqrt(xn2) + ptol));
        }
        // the candidate's H, g overwrite the current ones (an accepted step then needs no copy and no second Jacobian);
        // a rejected step -- rare -- restores them by re-evaluating at x
        double cost_c;
        const bool cand_ok = evaluate(xc, scale, H, g, cost_c);
        if constexpr (SPLIT == 1) {
            if (sx->timed_out) { failed = true; break; }
        }
        if (!cand_ok) cost_c = DBL_MAX;
        if (ptol_hit) {  // ParameterToleranceReached
            converged = true; trace(3, cost, cost_c, mcc, 0.0, step_norm); break;
        }
        const double cost_change = cost - cost_c;
        if (uniform(fabs(cost_change) <= ftol * cost)) {    // FunctionToleranceReached
            converged = true; trace(4, cost, cost_c, mcc, cost_change * fast_rcp(mcc), step_norm); break;
        }
        const double rel = cost_change * fast_rcp(mcc);
        if (uniform(rel > 1e-3)) {  // HandleSuccessfulStep
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = xc[j];
            const double cost_prev = cost;
            cost = cost_c;
            xn2 = sqnorm6(x);
            gmax = grad_max(g);
            const double tq = 2.0 * rel - 1.0;
            radius = fmin(1e16, radius * fast_rcp(fmax(1.0 / 3.0, 1.0 - tq * tq * tq)));
            dfac = 2.0;
            trace(1, cost_prev, cost_c, mcc, rel, step_norm);
        } else {
            radius /= dfac; dfac *= 2.0;
            trace(2, cost, cost_c, mcc, rel, step_norm);
            double cost_again;
            if (!evaluate(x, scale, H, g, cost_again)) { failed = true; break; }
        }
    }
    LC_PSTAMP(1);
#ifdef LC_STAMPS
    if (lane == 0 && p.iters) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.iters) + 8 * (size_t)b;
        for (int i = 0; i < 6; ++i) o[i] = pst_[i];
        o[6] = iter;
        o[7] = pst_[6];  // LDL^T solve alone (the rest of the LM algebra is in slot 1)
    }
#endif
    const bool invalid = failed || !converged;
    if (invalid && lane < 7) {
        const float v = fin(st_in[lane]);
        if (store && (p.start || start_override || filter)) p.states[7 * (size_t)b + lane] = v;
        if (handoff) handoff[lane] = v;
    }
    if (lane == 0) {
        if (store) {
            p.rets[b] = invalid ? 1 : 0;
            if constexpr (SPLIT == 1) {
                if (sx->timed_out) p.rets[b] = kPnpPartNeverArrived;  // never leaves the launch pair: the rescue launch re-solves the pose
            }
            p.result_tr[b] = (float)radius;
#ifndef LC_STAMPS
            if (p.iters) p.iters[b] = iter;
#endif
        }
        if (!invalid) {  // ceres.cpp:131-144: AngleAxisToQuaternion, write back in place
            float stq[7];
            float* st = stq;
            const double t2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
            double q0 = 1.0, kk = 0.5;
            if (t2 > 0.0) {
                const double th = fast_sqrt(t2), h = 0.5 * th;
                double sh, ch;
                sincos_small(h, sh, ch);  // < 1 ulp on the reduced interval, like the libm call it replaces
                q0 = ch;
                kk = sh / th;
            }
            st[0] = (float)q0; st[1] = (float)(x[0] * kk); st[2] = (float)(x[1] * kk); st[3] = (float)(x[2] * kk);
            st[4] = (float)x[3]; st[5] = (float)x[4]; st[6] = (float)x[5];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (store) p.states[7 * (size_t)b + k] = stq[k];
                if (handoff) handoff[k] = stq[k];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same solve for LARGE grids (B > kLatencyGridMax poses: several waves per SIMD, throughput matters, not one wave's latency)
// in at most 168 VGPRs, so that THREE waves fit a SIMD instead of two.  solve_pose keeps every wave-uniform quantity of the solve
// -- the 21 + 6 entries of the normal equations, pose, candidate, Jacobi scaling and its inverse: ~60 doubles = 120 VGPRs that
// hold the same value in all 64 lanes -- in registers; here they live in LDS (the block sum's 28 totals stay where the reduction
// leaves them, x / scale / 1/scale in a 24-double block behind them) and are read (ds_read_b64 of one address: a broadcast) where
// an expression needs them.  With three waves per SIMD the extra LDS latency is covered by the other waves.
// The arithmetic is solve_pose<true, 1>'s, expression for expression: same results bit for bit
// (tests/test_gpu_pnp.py::test_large_grid_build_equals_the_latency_build).
constexpr int kPnpLowregLdsDoubles = kPnpLdsDoubles<1> + 30;

// (A + diag(dg)) y = g with A (packed upper 21), g and the LM diagonal read from LDS; returns ok, y and the model cost change
__device__ __forceinline__ bool ldlt_solve6_lds(const double* A, const double* g, double inv_radius, double (&y)[6], double& mcc) {
    double L[6][6], Ld[6][6], id[6];
    bool ok = true;
    auto dgv = [&](int i) { return fmin(fmax(A[tri6(i, i)], 1e-6), 1e32) * inv_radius; };
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = A[tri6(j, j)] + dgv(j);
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * Ld[j][k];
        ok = ok && (dj > 0) && (dj < DBL_MAX);
        id[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = A[tri6(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * Ld[j][k];
            Ld[i][j] = v;
            L[i][j] = v * id[j];
        }
    }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = g[i];
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
    mcc = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) mcc += y[i] * (g[i] + dgv(i) * y[i]);
    mcc *= 0.5;
    return ok;
}

template <bool OPTS = false>
__device__ __forceinline__ void solve_pose_lowreg(const PnpParams& p, int b, int lane, double* bc) {
    const int n = p.counts ? p.counts[b] : p.Nmax;
    const int bk = (OPTS && p.pose_mod > 0) ? b % p.pose_mod : b;
    const float* st_in = (p.start ? p.start + 7 * (size_t)bk : p.states + 7 * (size_t)b);
    const bool filter = OPTS && (p.options & kPnpNanToNum);
    auto fin = [&](float f) { return filter ? nan_to_num(f) : f; };
    if (n < 3) {  // ceres.cpp:84-91
        if (lane == 0) {
            p.rets[b] = 1;
            p.result_tr[b] = 1.f;
            if (p.iters) p.iters[b] = 0;
        }
        if ((p.start || filter) && lane < 7) p.states[7 * (size_t)b + lane] = fin(st_in[lane]);
        return;
    }
    const size_t base = (size_t)b * p.Nmax;
    const bool active = lane < n;
    const RawPoint raw = load_raw_point<OPTS>(p, base, active ? lane : 0);
    double cam[6];
    {
        const float* Kp = p.K + 9 * (size_t)bk;
#pragma unroll
        for (int i = 0; i < 6; ++i) cam[i] = fin(Kp[i]);
    }
    // LDS: tot[0..20] = H, tot[21..26] = g, tot[27] = r.r of the last evaluation; us: x | scale | iscale | candidate
    double* const tot = block_sum_totals<28, 1>(bc);
    double* const ux = bc + kPnpLdsDoubles<1>;
    double* const uscale = ux + 6;
    double* const uiscale = ux + 12;
    double* const uxc = ux + 18;
    double* const ucam = ux + 24;  // K[0,0], K[0,1], -, K[1,0], K[1,1], - (the principal point is folded into the measurements)
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) ucam[i] = cam[i];
    }
    {
        double x[6];
        const double q0 = fin(st_in[0]), q1 = fin(st_in[1]), q2 = fin(st_in[2]), q3 = fin(st_in[3]);
        const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
        double kk = 2.0;
        if (s2 > 0.0) {
            const double s = fast_sqrt(s2);
            const double half = atan_ratio_pos(s, fabs(q0));
            const double two_theta = 2.0 * ((q0 < 0.0) ? -half : half);
            kk = two_theta / s;
        }
        x[0] = q1 * kk; x[1] = q2 * kk; x[2] = q3 * kk;
        x[3] = fin(st_in[4]); x[4] = fin(st_in[5]); x[5] = fin(st_in[6]);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) { ux[j] = x[j]; uscale[j] = 1.0; }
        }
        wave_sync();
    }
    Point rp = to_point(raw, cam);
    if (!active) { rp.a = 0.0; rp.b = 0.0; rp.c = 0.0; }

    // evaluation at xe (6 doubles in LDS) with the column scaling in uscale: the 28 totals stay in LDS; returns the cost, false when non-finite
    auto evaluate = [&](const double* xe, double& cost) -> bool {
        double xv[6], sc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { xv[j] = xe[j]; sc[j] = uscale[j]; }
        Rot rt;
        make_rot(xv, rt);
        const double t[3] = {xv[3], xv[4], xv[5]};
        double acc[28];
        const int pos = block_sum_open<1>(lane);
        accumulate_point<true, true, 1>(rp, rt, t, ucam, sc, acc, bc, pos);
        block_sum_reduce<28, 1>(bc, lane);
        const double ss = tot[27];
        cost = 0.5 * ss;
        double chk = ss;
#pragma unroll
        for (int i = 0; i < 6; ++i) chk += tot[tri6(i, i)];
        return chk <= DBL_MAX;
    };

    const double ftol = p.ftol, ptol = 1e-8, gtol = 1e-10;
    double cost;
    bool failed = !evaluate(ux, cost);
    {   // Jacobi scaling 1/(1+||J_j||), fixed at iteration 0; the totals become those of the scaled problem
        double scale[6], iscale[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            iscale[j] = 1.0 + fast_sqrt(tot[tri6(j, j)]);
            scale[j] = fast_rcp(iscale[j]);
        }
        double Hs[21], gs[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) { double h = tot[tri6(i, j)]; h *= scale[i] * scale[j]; Hs[tri6(i, j)] = h; }
            double gi = tot[21 + i]; gi *= scale[i]; gs[i] = gi;
        }
        wave_sync();  // every lane has read the unscaled totals
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 21; ++k) tot[k] = Hs[k];
#pragma unroll
            for (int j = 0; j < 6; ++j) { tot[21 + j] = gs[j]; uscale[j] = scale[j]; uiscale[j] = iscale[j]; }
        }
        wave_sync();
    }
    auto grad_max = [&]() {  // max-norm of the UNSCALED gradient J^T r
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m = fmax(m, fabs(tot[21 + j]) * uiscale[j]);
        return m;
    };
    auto sqnorm6 = [](const double* v) {
        double m = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) m += v[j] * v[j];
        return m;
    };
    double gmax = grad_max();
    double xn2 = sqnorm6(ux);
    double radius = 1e4, dfac = 2.0;
    int iter = 0, n_invalid = 0;
    bool converged = false;

    while (uniform(!failed && !converged)) {
        if (iter >= p.max_iter) break;
        if (uniform(gmax <= gtol || radius <= 1e-32)) { converged = true; break; }
        ++iter;
        double y[6], mcc;
        const double inv_radius = fast_rcp(radius);
        bool step_ok = ldlt_solve6_lds(tot, tot + 21, inv_radius, y, mcc);
        step_ok = step_ok && (mcc > 0.0) && (mcc <= DBL_MAX);
        if (uniform(!step_ok)) {  // HandleInvalidStep
            if (++n_invalid >= 5) { failed = true; break; }
            radius /= dfac; dfac *= 2.0;
            continue;
        }
        n_invalid = 0;
        double sn2;
        {
            double xc[6], delta[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { delta[j] = -y[j] * uscale[j]; xc[j] = ux[j] + delta[j]; }
            sn2 = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) sn2 += delta[j] * delta[j];
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) uxc[j] = xc[j];
            }
            wave_sync();
        }
        const bool ptol_in_reach = uniform(!(sn2 > 2.0 * ptol * ptol * (xn2 + ptol * ptol)));
        bool ptol_hit = false;
        if (ptol_in_reach) ptol_hit = uniform(fast_sqrt(sn2) <= ptol * (fast_sqrt(xn2) + ptol));
        double cost_c;
        const bool cand_ok = evaluate(uxc, cost_c);
        if (!cand_ok) cost_c = DBL_MAX;
        if (ptol_hit) { converged = true; break; }  // ParameterToleranceReached
        const double cost_change = cost - cost_c;
        if (uniform(fabs(cost_change) <= ftol * cost)) { converged = true; break; }  // FunctionToleranceReached
        const double rel = cost_change * fast_rcp(mcc);
        if (uniform(rel > 1e-3)) {  // HandleSuccessfulStep
            double xn[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) xn[j] = uxc[j];
            wave_sync();
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) ux[j] = xn[j];
            }
            wave_sync();
            cost = cost_c;
            xn2 = sqnorm6(ux);
            gmax = grad_max();
            const double tq = 2.0 * rel - 1.0;
            radius = fmin(1e16, radius * fast_rcp(fmax(1.0 / 3.0, 1.0 - tq * tq * tq)));
            dfac = 2.0;
        } else {
            radius /= dfac; dfac *= 2.0;
            double cost_again;
            if (!evaluate(ux, cost_again)) { failed = true; break; }
        }
    }
    const bool invalid = failed || !converged;
    if (invalid && (p.start || filter) && lane < 7) p.states[7 * (size_t)b + lane] = fin(st_in[lane]);
    if (lane == 0) {
        p.rets[b] = invalid ? 1 : 0;
        p.result_tr[b] = (float)radius;
        if (p.iters) p.iters[b] = iter;
        if (!invalid) {  // ceres.cpp:131-144: AngleAxisToQuaternion, write back in place
            float* st = p.states + 7 * (size_t)b;
            double x[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = ux[j];
            const double t2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
            double q0 = 1.0, kk = 0.5;
            if (t2 > 0.0) {
                const double th = fast_sqrt(t2), h = 0.5 * th;
                double sh, ch;
                sincos_small(h, sh, ch);
                q0 = ch;
                kk = sh / th;
            }
            st[0] = (float)q0; st[1] = (float)(x[0] * kk); st[2] = (float)(x[1] * kk); st[3] = (float)(x[2] * kk);
            st[4] = (float)x[3]; st[5] = (float)x[4]; st[6] = (float)x[5];
        }
    }
}

}  // namespace pnp
}  // namespace lc
