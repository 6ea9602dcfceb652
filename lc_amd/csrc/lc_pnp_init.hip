// GPU initialiser for the weighted PnP solve: RANSAC over minimal P3P hypotheses.
//
// Takes the place of lib/pnp/cv2_solver.py:69-88 (cv2.solvePnPRansac, EPnP kernel, 150 iterations, multiprocessing.Pool)
// in front of cer_solver.solve (test.py:59,120): one hypothesis per lane (64 x rounds hypotheses >= the reference's 150), every
// hypothesis is scored against the correspondences, the one with most inliers wins (ties: smaller inlier error, then the smaller
// index) and the pose + the inlier mask are written.  No device->host->device trip, no process pool.  Two launch forms with the same
// hypothesis stream, per-point arithmetic and ordering:
//   * lc_pnp_ransac_kernel: one workgroup per pose, its rounds on separate wavefronts, points staged in LDS -- for batches that fill
//     the chip with one workgroup per pose (or few points);
//   * lc_ransac_{hypotheses,score,select}_kernel: three launches over a workspace that spread the points of one pose over the chip
//     (further down; lc_ransac_score_select_kernel: scoring and selection as one launch, an option that measured slower).
// Either form can also write the 'weighted-filtered' re-selection of test.py:129-133 (the winner's inliers compacted to the front of
// their rows: lc_pnp_ransac_init4_f32), which the workgroup that writes the inlier mask does on the way.
// OpenCV's RNG/EPnP cannot be reproduced bit for bit (and OpenCV is absent here: parity at this boundary is unpinned and outside the
// metric); the contract kept is the role: a pose inside the LM basin of convergence plus an inlier set for
// `weighted_filtered` (test.py:129-134); the kernels' own contract is pinned by oracle/p3p_ransac_oracle.py.
//
// P3P: depths l_i of three bearings y_i with |l_i y_i - l_j y_j|^2 = |x_i - x_j|^2.  The pencil D1 + g D2 of the two
// constant-free quadrics is made singular by a root g of a cubic (coefficients from 3x3 determinants), the singular quadric
// splits into two planes through its eigen-decomposition, each plane cuts D1 in <= 2 rays (a quadratic), the scale comes
// from one distance constraint; the candidate that reprojects a fourth correspondence best is Gauss-Newton polished and R,t follow
// from the three point pairs.  ONE root is decomposed and its up-to-four candidates are formed without branches (P3P::solve).  (Divisions and square roots are the Newton-refined v_rcp_f64 / v_rsq_f64 forms of lc_common.h,
// 2-4e-15 relative: the result is polished and compared at 1e-4.)
#include <algorithm>
#include <cfloat>

#include "lc_common.h"
#include "lc_kernels.h"
#include "lc_select_rows.h"

#ifdef LC_P3P_STAMPS
namespace lc { namespace p3p_diag { __device__ unsigned long long g_p3p_stamp[7]; __device__ unsigned long long g_sel_stamp[8]; } }
#endif

namespace lc {
namespace {

#ifdef LC_P3P_STAMPS  // diagnostic build (scripts/ubench/p3p_stamps.py): shader-clock stamps of the hypothesis kernel's phases
#define LC_P3P_STAMP(i) ::lc::p3p_diag::g_p3p_stamp[i] = __builtin_amdgcn_s_memtime()
#define LC_SEL_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) ::lc::p3p_diag::g_sel_stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define LC_P3P_STAMP(i)
#define LC_SEL_STAMP(i)
#endif

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Sym3 { double a00, a01, a02, a11, a12, a22; };
__device__ __forceinline__ double det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
__device__ __forceinline__ V3 row(const Sym3& m, int r) {
    return r == 0 ? V3{m.a00, m.a01, m.a02} : (r == 1 ? V3{m.a01, m.a11, m.a12} : V3{m.a02, m.a12, m.a22});
}
// unit null vector of a singular symmetric matrix: the largest cross product of two rows
__device__ __forceinline__ V3 null_vec(const Sym3& m) {
    const V3 c01 = cross(row(m, 0), row(m, 1)), c02 = cross(row(m, 0), row(m, 2)), c12 = cross(row(m, 1), row(m, 2));
    const double n01 = dot(c01, c01), n02 = dot(c02, c02), n12 = dot(c12, c12);
    V3 v = c01;
    double n = n01;
    if (n02 > n) { v = c02; n = n02; }
    if (n12 > n) { v = c12; n = n12; }
    double sn, rn;
    fast_sqrt_rsqrt(fmax(n, DBL_MIN), sn, rn);
    return rn * v;
}

struct Pose { double R[9], t[3]; };

// Grunert-type P3P through the singular member of the pencil of the two constant-free quadrics.  solve() hands every candidate's
// three depths to emit(l) (no array of solutions: a dynamically indexed Pose[4] lives in scratch memory); the caller chooses
// one, polish() refines its depths by Gauss-Newton on the three distance constraints, pose() turns depths into R, t.
struct P3P {
    V3 y[3], x[3];
    double c12, c13, c23, a12, a13, a23;
    double Xi[9];  // X^-1, X = [x1 - x2, x1 - x3, their cross product] (columns)

    __device__ __forceinline__ bool init(const V3 (&y_)[3], const V3 (&x_)[3]) {
        for (int k = 0; k < 3; ++k) { y[k] = y_[k]; x[k] = x_[k]; }
        c12 = dot(y[0], y[1]); c13 = dot(y[0], y[2]); c23 = dot(y[1], y[2]);
        const V3 d12 = x[0] - x[1], d13 = x[0] - x[2], d23 = x[1] - x[2];
        a12 = dot(d12, d12); a13 = dot(d13, d13); a23 = dot(d23, d23);
        const V3 dn = cross(d12, d13);
        if (!(dot(dn, dn) > 1e-12 * a12 * a13)) return false;  // collinear sample
        const double X[9] = {d12.x, d13.x, dn.x, d12.y, d13.y, dn.y, d12.z, d13.z, dn.z};
        const double dt = det3(X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], X[8]);
        const double id = fast_rcp(dt);
        Xi[0] = (X[4] * X[8] - X[5] * X[7]) * id; Xi[1] = (X[2] * X[7] - X[1] * X[8]) * id; Xi[2] = (X[1] * X[5] - X[2] * X[4]) * id;
        Xi[3] = (X[5] * X[6] - X[3] * X[8]) * id; Xi[4] = (X[0] * X[8] - X[2] * X[6]) * id; Xi[5] = (X[2] * X[3] - X[0] * X[5]) * id;
        Xi[6] = (X[3] * X[7] - X[4] * X[6]) * id; Xi[7] = (X[1] * X[6] - X[0] * X[7]) * id; Xi[8] = (X[0] * X[4] - X[1] * X[3]) * id;
        return true;
    }

    // R = [z1 - z2, z1 - z3, cross] X^-1 with z_i = l_i y_i, t = z1 - R x1
    __device__ __forceinline__ void pose(const double (&l)[3], Pose& o) const {
        const V3 z1 = l[0] * y[0], z2 = l[1] * y[1], z3 = l[2] * y[2];
        const V3 yd1 = z1 - z2, yd2 = z1 - z3, yn = cross(yd1, yd2);
        const double Y[9] = {yd1.x, yd2.x, yn.x, yd1.y, yd2.y, yn.y, yd1.z, yd2.z, yn.z};
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) o.R[3 * a + c] = Y[3 * a] * Xi[c] + Y[3 * a + 1] * Xi[3 + c] + Y[3 * a + 2] * Xi[6 + c];
        const V3 Rx = {o.R[0] * x[0].x + o.R[1] * x[0].y + o.R[2] * x[0].z, o.R[3] * x[0].x + o.R[4] * x[0].y + o.R[5] * x[0].z,
                       o.R[6] * x[0].x + o.R[7] * x[0].y + o.R[8] * x[0].z};
        o.t[0] = z1.x - Rx.x; o.t[1] = z1.y - Rx.y; o.t[2] = z1.z - Rx.z;
    }

    // three Gauss-Newton steps on |l_i y_i - l_j y_j|^2 = a_ij (Cramer's rule on the Jacobian [[J00,J01,0],[J10,0,J12],[0,J21,J22]],
    // zeros folded by hand: the compiler may not drop 0 * x); false when a depth ends non-positive
    __device__ __forceinline__ bool polish(double (&l)[3]) const {
        for (int it = 0; it < 3; ++it) {
            const double r0 = l[0] * l[0] + l[1] * l[1] - 2.0 * c12 * l[0] * l[1] - a12;
            const double r1 = l[0] * l[0] + l[2] * l[2] - 2.0 * c13 * l[0] * l[2] - a13;
            const double r2 = l[1] * l[1] + l[2] * l[2] - 2.0 * c23 * l[1] * l[2] - a23;
            const double J00 = 2.0 * (l[0] - c12 * l[1]), J01 = 2.0 * (l[1] - c12 * l[0]);
            const double J10 = 2.0 * (l[0] - c13 * l[2]), J12 = 2.0 * (l[2] - c13 * l[0]);
            const double J21 = 2.0 * (l[1] - c23 * l[2]), J22 = 2.0 * (l[2] - c23 * l[1]);
            const double dt = -J00 * J12 * J21 - J01 * J10 * J22;
            if (!(fabs(dt) > 1e-300)) break;
            const double id = fast_rcp(dt);
            const double m = r1 * J22 - J12 * r2;
            l[0] -= (-r0 * J12 * J21 - J01 * m) * id;
            l[1] -= (J00 * m - r0 * J10 * J22) * id;
            l[2] -= (r0 * J10 * J21 - J00 * r1 * J21 - J01 * J10 * r2) * id;
        }
        return l[0] > 0 && l[1] > 0 && l[2] > 0;
    }

    // Up to four candidates -> L[c] (three depths), ok[c]; c = 2 * (plane of the pair) + (root of that plane's quadratic).
    // ONE root of the cubic is decomposed: every real solution of the two quadrics lies on the plane pair of any root whose singular
    // quadric splits into real planes (Persson & Nordberg's Lambda Twist takes one root for the same reason); a root without real
    // planes is skipped by a cheap scan.  The four candidates are computed WITHOUT branches -- a lane's invalid ones carry a false
    // flag and whatever arithmetic produced them -- so that their four dependency chains interleave in the one wavefront a SIMD
    // runs here (the branching form evaluated them one after the other: 7.3 k of the kernel's 13.9 k cycles).
    __device__ __forceinline__ void solve(double (&L)[4][3], bool (&ok)[4]) const {
        // D1 = a23 M12 - a12 M23,  D2 = a23 M13 - a13 M23
        const Sym3 D1{a23, -a23 * c12, 0.0, a23 - a12, a12 * c23, -a12};
        const Sym3 D2{a23, 0.0, -a23 * c13, -a13, a13 * c23, a23 - a13};
        // det(D1 + g D2) = k0 + k1 g + k2 g^2 + k3 g^3 by multilinearity in the columns
        auto col = [](const Sym3& m, int c) { return row(m, c); };
        auto det_cols = [](V3 p, V3 q, V3 r) { return det3(p.x, q.x, r.x, p.y, q.y, r.y, p.z, q.z, r.z); };
        const V3 p0 = col(D1, 0), p1 = col(D1, 1), p2 = col(D1, 2), q0 = col(D2, 0), q1 = col(D2, 1), q2 = col(D2, 2);
        const double k0 = det_cols(p0, p1, p2);
        const double k1 = det_cols(q0, p1, p2) + det_cols(p0, q1, p2) + det_cols(p0, p1, q2);
        const double k2 = det_cols(p0, q1, q2) + det_cols(q0, p1, q2) + det_cols(q0, q1, p2);
        const double k3 = det_cols(q0, q1, q2);
        LC_P3P_STAMP(2);
        // real roots of the cubic (trigonometric / Cardano on the depressed form), Newton-polished
        double roots[3] = {0.0, 0.0, 0.0};
        int nroots = 0;
        if (fabs(k3) > 1e-14 * (fabs(k0) + fabs(k1) + fabs(k2) + fabs(k3))) {
            const double ik3 = fast_rcp(k3);
            const double b = k2 * ik3, c = k1 * ik3, d = k0 * ik3;
            const double p = c - b * b / 3.0, q = 2.0 * b * b * b / 27.0 - b * c / 3.0 + d;
            const double disc = q * q / 4.0 + p * p * p / 27.0;
            if (disc > 0) {
                const double s = fast_sqrt(disc);
                roots[nroots++] = cbrt(-q / 2.0 + s) + cbrt(-q / 2.0 - s) - b / 3.0;
            } else {
                const double m = 2.0 * fast_sqrt(fmax(-p / 3.0, 0.0));
                const double arg = m > 0 ? fmin(1.0, fmax(-1.0, 3.0 * q * fast_rcp(p * m))) : 0.0;
                const double th = acos(arg) / 3.0;
                for (int k = 0; k < 3; ++k) roots[nroots++] = m * cos(th - 2.0943951023931953 * k) - b / 3.0;
            }
            for (int r = 0; r < nroots; ++r) {
                double g = roots[r];
                for (int it = 0; it < 3; ++it) {
                    const double f = ((g + b) * g + c) * g + d, fp = (3.0 * g + 2.0 * b) * g + c;
                    if (fabs(fp) > 0) g -= f * fast_rcp(fp);
                }
                roots[r] = g;
            }
        } else if (fabs(k2) > 0) {  // degenerate cubic: quadratic
            const double disc = k1 * k1 - 4.0 * k2 * k0;
            if (disc >= 0) {
                const double s = sqrt(disc);
                roots[nroots++] = (-k1 + s) / (2.0 * k2);
                roots[nroots++] = (-k1 - s) / (2.0 * k2);
            }
        }
        LC_P3P_STAMP(3);
        // the first root whose singular quadric has a real plane pair (18 flops per root)
        Sym3 D0{0, 0, 0, 0, 0, 0};
        double tr = 0, m2 = 0;
        bool found = false;
        for (int r = 0; r < nroots && !found; ++r) {
            const double g = r == 0 ? roots[0] : (r == 1 ? roots[1] : roots[2]);
            D0 = Sym3{D1.a00 + g * D2.a00, D1.a01 + g * D2.a01, D1.a02 + g * D2.a02, D1.a11 + g * D2.a11, D1.a12 + g * D2.a12,
                      D1.a22 + g * D2.a22};
            // eigenvalues s1, s2 of the rank-2 matrix: s1 + s2 = trace, s1 s2 = sum of principal 2x2 minors
            tr = D0.a00 + D0.a11 + D0.a22;
            m2 = D0.a00 * D0.a11 - D0.a01 * D0.a01 + D0.a00 * D0.a22 - D0.a02 * D0.a02 + D0.a11 * D0.a22 - D0.a12 * D0.a12;
            found = m2 < 0;  // needs eigenvalues of opposite sign to split into two real planes
        }
        const double sq = fast_sqrt(fmax(tr * tr - 4.0 * m2, 0.0));
        const double s1 = 0.5 * (tr + sq), s2 = 0.5 * (tr - sq);  // s1 > 0 > s2
        const Sym3 S1{D0.a00 - s1, D0.a01, D0.a02, D0.a11 - s1, D0.a12, D0.a22 - s1};
        const V3 e1 = null_vec(S1), e0 = null_vec(D0);
        const V3 e2 = cross(e0, e1);
        const double sgm = fast_sqrt(fmax(-s2 * fast_rcp(s1), 0.0));
#pragma unroll
        for (int sign = 0; sign < 2; ++sign) {
            const V3 pl = e1 + ((sign ? -sgm : sgm) * e2);  // plane pl . lambda = 0
            // eliminate the component with the largest |pl|: lambda_k = u lambda_i + v lambda_j, (i, j, k) = (1,2,0) | (0,2,1) | (0,1,2)
            const double ax = fabs(pl.x), ay = fabs(pl.y), az = fabs(pl.z);
            const bool k0_ = ax >= ay && ax >= az, k1_ = !k0_ && ay >= az;
            const double pk = k0_ ? pl.x : (k1_ ? pl.y : pl.z), pi = k0_ ? pl.y : pl.x, pj = (k0_ || k1_) ? pl.z : pl.y;
            const double ipk = fast_rcp(pk);
            const double u = -pi * ipk, v = -pj * ipk;
            const double Dii = k0_ ? D1.a11 : D1.a00, Djj = (k0_ || k1_) ? D1.a22 : D1.a11, Dkk = k0_ ? D1.a00 : (k1_ ? D1.a11 : D1.a22);
            const double Dij = k0_ ? D1.a12 : (k1_ ? D1.a02 : D1.a01), Dik = k0_ ? D1.a01 : (k1_ ? D1.a01 : D1.a02),
                         Djk = k0_ ? D1.a02 : D1.a12;
            const double A = Dii + Dkk * u * u + 2.0 * Dik * u;
            const double C = Djj + Dkk * v * v + 2.0 * Djk * v;
            const double Bq = 2.0 * (Dkk * u * v + Dij + Dik * v + Djk * u);
            // A + Bq tau + C tau^2 = 0, tau = lambda_j / lambda_i
            const bool quad = fabs(C) > 1e-14 * (fabs(A) + fabs(Bq) + fabs(C));
            const double disc = Bq * Bq - 4.0 * A * C;
            const double s = fast_sqrt(fmax(disc, 0.0)), qq = -0.5 * (Bq + (Bq >= 0 ? s : -s));
            const double tau[2] = {quad ? qq * fast_rcp(C) : -A * fast_rcp(Bq), A * fast_rcp(qq)};
            const bool have[2] = {found && (quad ? disc >= 0 : fabs(Bq) > 0), found && quad && disc >= 0 && qq != 0};
            const double cij = k0_ ? c23 : (k1_ ? c13 : c12), aij = k0_ ? a23 : (k1_ ? a13 : a12);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const double den = 1.0 + tau[tt] * tau[tt] - 2.0 * cij * tau[tt];
                const double li = fast_sqrt(fmax(aij * fast_rcp(den), 0.0)), lj = tau[tt] * li, lk = u * li + v * lj;
                const int c = 2 * sign + tt;
                ok[c] = have[tt] && tau[tt] > 0 && den > 0 && lk > 0;
                L[c][0] = k0_ ? lk : li;
                L[c][1] = k0_ ? li : (k1_ ? lk : lj);
                L[c][2] = (k0_ || k1_) ? lj : lk;
            }
        }
        LC_P3P_STAMP(4);
    }
};


// rotation matrix -> quaternion (w,x,y,z), w >= 0
__device__ void mat_to_quat(const double R[9], float q[4]) {
    const double tr = R[0] + R[4] + R[8];
    double w, x, y, z;
    if (tr > 0) {
        double s, is;
        fast_sqrt_rsqrt(tr + 1.0, s, is);
        s *= 2.0; is *= 0.5;
        w = 0.25 * s; x = (R[7] - R[5]) * is; y = (R[2] - R[6]) * is; z = (R[3] - R[1]) * is;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s, is;
        fast_sqrt_rsqrt(1.0 + R[0] - R[4] - R[8], s, is);
        s *= 2.0; is *= 0.5;
        w = (R[7] - R[5]) * is; x = 0.25 * s; y = (R[1] + R[3]) * is; z = (R[2] + R[6]) * is;
    } else if (R[4] > R[8]) {
        double s, is;
        fast_sqrt_rsqrt(1.0 + R[4] - R[0] - R[8], s, is);
        s *= 2.0; is *= 0.5;
        w = (R[2] - R[6]) * is; x = (R[1] + R[3]) * is; y = 0.25 * s; z = (R[5] + R[7]) * is;
    } else {
        double s, is;
        fast_sqrt_rsqrt(1.0 + R[8] - R[0] - R[4], s, is);
        s *= 2.0; is *= 0.5;
        w = (R[3] - R[1]) * is; x = (R[2] + R[6]) * is; y = (R[5] + R[7]) * is; z = 0.25 * s;
    }
    double nn, n;
    fast_sqrt_rsqrt(w * w + x * x + y * y + z * z, nn, n);
    const double sg = w < 0 ? -n : n;
    q[0] = (float)(w * sg); q[1] = (float)(x * sg); q[2] = (float)(y * sg); q[3] = (float)(z * sg);
}

// Points of a pose the single-launch form keeps in LDS at a time (40 KB); a pose with more is scored tile after tile.  EVERY point of
// a pose takes part in the sampling and in the scoring, as in cv2.solvePnPRansac (cv2_solver.py:72-75) -- rounds 1-3 stopped at the
// first 2048 of a row, which at zlmo's test-time shape (128x128 candidates, selection compacted in raster order) was the upper rows
// of the object only.
constexpr int kLdsTilePts = 2048;

// Hypothesis `hyp` of pose b: four distinct sample indices below nl from a counter-based hash stream
// (oracle/p3p_ransac_oracle.py:sample_indices restates it bit for bit).
__device__ __forceinline__ void sample_indices(unsigned seed, int b, int hyp, int nl, int (&idx)[4]) {
    unsigned h = hash_u32(seed ^ hash_u32((unsigned)b * 0x9E3779B9u + (unsigned)hyp));
    for (int k = 0; k < 4; ++k) {
        h = hash_u32(h + 0x6D2B79F5u);
        int v = (int)(h % (unsigned)nl);
        for (int guard = 0; guard < 8; ++guard) {
            bool dup = false;
            for (int m = 0; m < k; ++m) dup = dup || (idx[m] == v);
            if (!dup) break;
            v = (v + 1) % nl;
        }
        idx[k] = v;
    }
}

// Minimal-sample pose: P3P on the first three correspondences, the fourth disambiguates the up-to-4 solutions (best reprojection).
// point(i, X, u): 3D point and normalised image coordinates K^-1 (u,v,1) of correspondence i as floats.  False: no usable solution.
template <class PointFn>
__device__ __forceinline__ bool hypothesis_pose(const int (&idx)[4], PointFn&& point, Pose& out) {
    V3 y[3], x[3];
    for (int k = 0; k < 3; ++k) {
        float X[3], u[2];
        point(idx[k], X, u);
        const double ux = u[0], uy = u[1];
        double nrm, inv;
        fast_sqrt_rsqrt(ux * ux + uy * uy + 1.0, nrm, inv);
        y[k] = {ux * inv, uy * inv, inv};
        x[k] = {X[0], X[1], X[2]};
    }
    P3P geo;
    if (!geo.init(y, x)) return false;
    float X4[3], u4[2];
    point(idx[3], X4, u4);
    bool have = false;
    float pick_e = INFINITY;
    double lbest[3] = {0.0, 0.0, 0.0};
    // The candidate that reprojects the 4th point best (first one on ties) is chosen on its closed-form depths; only the chosen one is
    // Gauss-Newton polished (the closed form is good to ~1e-10, far below the gap between two P3P solutions; polishing all four
    // candidates before the choice was half of this kernel's instructions).
    double L[4][3];
    bool ok[4];
    geo.solve(L, ok);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Pose sol;
        geo.pose(L[c], sol);
        const float X = X4[0], Y = X4[1], Z = X4[2];
        const float cx = (float)sol.R[0] * X + (float)sol.R[1] * Y + (float)sol.R[2] * Z + (float)sol.t[0];
        const float cy = (float)sol.R[3] * X + (float)sol.R[4] * Y + (float)sol.R[5] * Z + (float)sol.t[1];
        const float cz = (float)sol.R[6] * X + (float)sol.R[7] * Y + (float)sol.R[8] * Z + (float)sol.t[2];
        const float ex = cx / cz - u4[0], ey = cy / cz - u4[1];
        const float e = ex * ex + ey * ey;
        if (ok[c] && cz > 0 && e < pick_e) { pick_e = e; lbest[0] = L[c][0]; lbest[1] = L[c][1]; lbest[2] = L[c][2]; have = true; }
    }
    if (!have || !geo.polish(lbest)) return false;
    geo.pose(lbest, out);
    return true;
}

// One correspondence against one hypothesis, in fp32 and WITHOUT a division, so that every bit is reproducible off the chip
// (oracle/p3p_ransac_oracle.py: score_f32 restates these lines operation by operation; v_rcp_f32's last bit cannot be restated):
//   c = R X + t by three fma chains that start from t;   r = (cx - u cz, cy - v cz);   q = ry ry + rx rx   (one fma over one product);
//   inlier  <=>  cz > 0  and  q < (thr2 cz) cz                  -- | c_xy / cz - u |^2 < thr2 with both sides multiplied by cz^2;
//   the inlier error of a hypothesis (the tie-break of equal counts) adds q over its inliers -- even / odd points of a 64-point chunk in
//   two sums, (even + odd) x 1 / (tz tz) per chunk (chunk_error: squared residuals in the camera plane at the hypothesis' own depth tz; +inf for tz <= 0;
//   IEEE operations only), chunk values in chunk order.  Every launch form performs exactly these operations in exactly this order.
__device__ __forceinline__ bool inlier_q(const float (&R)[9], const float (&t)[3], float X, float Y, float Z, float u, float v, float thr2, float& q) {
    const float cz = __builtin_fmaf(R[8], Z, __builtin_fmaf(R[7], Y, __builtin_fmaf(R[6], X, t[2])));
    const float cx = __builtin_fmaf(R[2], Z, __builtin_fmaf(R[1], Y, __builtin_fmaf(R[0], X, t[0])));
    const float cy = __builtin_fmaf(R[5], Z, __builtin_fmaf(R[4], Y, __builtin_fmaf(R[3], X, t[1])));
    const float rx = __builtin_fmaf(-u, cz, cx), ry = __builtin_fmaf(-v, cz, cy);
    q = __builtin_fmaf(ry, ry, rx * rx);
    return cz > 0 && q < (thr2 * cz) * cz;
}
__device__ __forceinline__ void score_point(const float (&R)[9], const float (&t)[3], float X, float Y, float Z, float u, float v,
                                            float thr2, int& cnt, float& err) {
    float q;
    const bool in = inlier_q(R, t, X, Y, Z, u, v, thr2, q);
    cnt += in ? 1 : 0;
    err += in ? q : 0.f;
}
// (even + odd) sums of a chunk -> the chunk's contribution to the hypothesis' inlier error
// (a hypothesis whose origin lies behind the camera, tz <= 0, has no depth to scale by: it ranks LAST among equal counts, error = +inf,
// instead of mixing an unscaled sum into the ordering)
__device__ __forceinline__ float chunk_error(float even, float odd, float tz) {
    if (!(tz > 0.f)) return __builtin_inff();
    const float scale = 1.f / (tz * tz);  // IEEE division (hipcc's default for fp32), once per hypothesis and chunk
    return (even + odd) * scale;
}

// two correspondences per instruction (v_pk_fma_f32 / v_pk_mul_f32): X, Y, Z, nu = -u, nv = -v hold the same coordinate of points i and i+1
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t bc2(float a) { return v2f_t{a, a}; }
__device__ __forceinline__ void score_pair(const float (&R)[9], const float (&t)[3], v2f_t X, v2f_t Y, v2f_t Z, v2f_t nu, v2f_t nv,
                                           float thr2, int& cnt, v2f_t& err) {
    const v2f_t cz = __builtin_elementwise_fma(bc2(R[8]), Z, __builtin_elementwise_fma(bc2(R[7]), Y, __builtin_elementwise_fma(bc2(R[6]), X, bc2(t[2]))));
    const v2f_t cx = __builtin_elementwise_fma(bc2(R[2]), Z, __builtin_elementwise_fma(bc2(R[1]), Y, __builtin_elementwise_fma(bc2(R[0]), X, bc2(t[0]))));
    const v2f_t cy = __builtin_elementwise_fma(bc2(R[5]), Z, __builtin_elementwise_fma(bc2(R[4]), Y, __builtin_elementwise_fma(bc2(R[3]), X, bc2(t[1]))));
    const v2f_t rx = __builtin_elementwise_fma(nu, cz, cx), ry = __builtin_elementwise_fma(nv, cz, cy);
    const v2f_t q = __builtin_elementwise_fma(ry, ry, rx * rx);
    const v2f_t lim = (bc2(thr2) * cz) * cz;
    const bool in0 = cz.x > 0 && q.x < lim.x, in1 = cz.y > 0 && q.y < lim.y;
    cnt += (in0 ? 1 : 0) + (in1 ? 1 : 0);
    err += v2f_t{in0 ? q.x : 0.f, in1 ? q.y : 0.f};
}

// (count, -error, -hypothesis id) ordering of the hypotheses of a pose
__device__ __forceinline__ bool better_hyp(int oc, float oe, int oh, int c, float e, int h) {
    return oc > c || (oc == c && (oe < e || (oe == e && oh < h)));
}


// K^-1 of the upper-triangular-free 2x3 camera block: u = k0 X/Z + k1 Y/Z + k2, v = k3 X/Z + k4 Y/Z + k5 (row 2 of K is (0,0,1) as in
// ceres.cpp:46-47); normalised coordinates of a pixel, rounded to float once
struct CamInv {
    double k0, k1, k2, k3, k4, k5, idet;
    __device__ explicit CamInv(const float* Kp) : k0(Kp[0]), k1(Kp[1]), k2(Kp[2]), k3(Kp[3]), k4(Kp[4]), k5(Kp[5]) { idet = 1.0 / (k0 * k4 - k1 * k3); }
    __device__ __forceinline__ void normalise(float px, float py, float& ux, float& uy) const {
        const double du = (double)px - k2, dv = (double)py - k5;
        ux = (float)((k4 * du - k1 * dv) * idet);
        uy = (float)((-k3 * du + k0 * dv) * idet);
    }
};

// Inlier threshold of pose b in pixels: the scalar; a per-pose value; or (per_pose_divides, lc_pnp_ransac_init5_f32 only) scalar / per-pose
// value -- test.py:56-57,115-116's `2 / gt_dict['out_pix_scale']` (rel_reproj_err) formed here instead of by two element-wise launches in front
// of the pipeline (IEEE division: the float torch's `2 / t` produces).  A divisor that is not positive leaves the scalar (an infinite
// threshold would call every point an inlier).
__device__ __forceinline__ float threshold_px(const RansacParams& p, int b) {
    if (!p.reproj_err_per_pose) return p.reproj_err;
    const float v = p.reproj_err_per_pose[b];
    if (!p.per_pose_divides) return v;
    return v > 0.f ? p.reproj_err / v : p.reproj_err;
}

__device__ __forceinline__ RowCopy selection_rows(const RansacParams& p) {
    return RowCopy{p.pts2d, p.sel_w, p.pts3d, p.sel_in_index, p.sel_pts2d, p.sel_w_out, p.sel_pts3d, p.sel_index, 0};
}

// The correspondences (and, for the re-selection, their weights and source indices) of kBatch consecutive chunks of blockDim.x
// entries, requested together: one memory round trip per batch.  cap: entries below it exist (rows are padded to Nmax, so a caller
// that does not know the pose's count yet asks with cap = Nmax and applies the count later).
constexpr int kBatch = 4;
struct PointBatch {
    float X[kBatch], Y[kBatch], Z[kBatch], pu[kBatch], pv[kBatch];
    float2 sw[kBatch];
    int src[kBatch];
};
__device__ __forceinline__ PointBatch load_batch(const RansacParams& p, size_t base, int i0, int cap) {
    // Every request of the batch is issued before the first value is used: the index is clamped instead of guarded (a slot behind the row's
    // end re-reads its last point and is zeroed afterwards) and the optional rows are read through pointers that fall back to rows that
    // exist.  Written as `have ? load : 0` per element, each of the kBatch slots was a branch with its loads and an s_waitcnt vmcnt(0):
    // four dependent memory round trips at the head of the selection kernel (ISA, round 6).
    const bool sel = p.sel_w != nullptr, idx = sel && p.sel_in_index != nullptr;
    const float* const sw = sel ? p.sel_w : p.pts2d;                                        // (B,Nmax,2) either way
    const int* const si = idx ? p.sel_in_index : reinterpret_cast<const int*>(p.pts2d);     // any readable (B,Nmax) words
    PointBatch q;
    if (cap <= 0) {  // (uniform) an empty row: nothing may be read
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            q.X[k] = q.Y[k] = q.Z[k] = q.pu[k] = q.pv[k] = 0.f;
            q.sw[k] = make_float2(0.f, 0.f);
            q.src[k] = i0 + k * (int)blockDim.x + (int)threadIdx.x;
        }
        return q;
    }
    float2 u[kBatch];
    int s[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        const int i = i0 + k * (int)blockDim.x + (int)threadIdx.x, j = i < cap ? i : cap - 1;
        const float* X = p.pts3d + (base + j) * 3;
        q.X[k] = X[0]; q.Y[k] = X[1]; q.Z[k] = X[2];
        u[k] = *reinterpret_cast<const float2*>(p.pts2d + (base + j) * 2);
        q.sw[k] = *reinterpret_cast<const float2*>(sw + (base + j) * 2);
        s[k] = si[base + j];
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        const int i = i0 + k * (int)blockDim.x + (int)threadIdx.x;
        const bool have = i < cap;
        q.X[k] = have ? q.X[k] : 0.f; q.Y[k] = have ? q.Y[k] : 0.f; q.Z[k] = have ? q.Z[k] : 0.f;
        q.pu[k] = have ? u[k].x : 0.f; q.pv[k] = have ? u[k].y : 0.f;
        q.sw[k] = (have && sel) ? q.sw[k] : make_float2(0.f, 0.f);
        q.src[k] = (have && idx) ? s[k] : i;
    }
    return q;
}

// mask[from, to) := 0 by the calling workgroup: 16-byte stores between the aligned ends (rows of 16 384 candidates: one byte per thread and
// iteration was 64 iterations of the selection kernel -- most of its 27 us at zlmo's test-time shape)
__device__ __forceinline__ void zero_bytes(unsigned char* m, int from, int to) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (to <= from) return;
    const uintptr_t a = reinterpret_cast<uintptr_t>(m);
    int lo = from + (int)((16 - ((a + from) & 15)) & 15);  // first 16-byte boundary at or after `from`
    if (lo > to) lo = to;
    const int hi = lo + ((to - lo) & ~15);
    for (int i = from + tid; i < lo; i += nthr) m[i] = 0;
    for (int i = lo + 16 * tid; i < hi; i += 16 * nthr) *reinterpret_cast<uint4*>(m + i) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = hi + tid; i < to; i += nthr) m[i] = 0;
}

// Winner of pose b -> outputs: inlier mask over ALL n points, their count, the pose as quaternion + translation, flags and (when
// asked for) the inliers compacted to the front of the selection rows.  Called by every thread of the workgroup; cc: LDS scratch, one
// int per (chunk of the batch, wavefront); first: load_batch(p, base, 0, cap >= n), which the caller may have requested long before
// the winner was known.
constexpr int kSelMaxWaves = 16;  // wavefronts of the workgroup that writes a pose's result (4; 16 for rows of more than 4096 candidates)
typedef int ChunkCounts[kBatch][kSelMaxWaves];
__device__ __forceinline__ void write_result(const RansacParams& p, int b, int n, bool ok, const double* bp, int win_hyp, float thr2,
                                             const CamInv& kin, ChunkCounts& cc, const PointBatch& first) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    const size_t base = (size_t)b * p.Nmax;
    unsigned char* mask = p.inlier_mask + base;
    const bool sel = p.sel_w != nullptr;
    const RowCopy rows = selection_rows(p);
    if (tid == nthr - 1) {  // everything but the inlier count is known: written now, under the loop's memory traffic
        float* st = p.states + 7 * (size_t)b;
        if (ok) {
            mat_to_quat(bp, st);
            st[4] = (float)bp[9]; st[5] = (float)bp[10]; st[6] = (float)bp[11];
        } else {
            st[0] = 1; st[1] = st[2] = st[3] = st[4] = st[5] = st[6] = 0;
            p.n_inliers[b] = 0;
        }
        p.invalid[b] = ok ? 0 : 1;
        if (p.best_hyp) p.best_hyp[b] = ok ? win_hyp : -1;
        if (p.valid_counts) p.valid_counts[b] = ok ? n : 0;
    }
    int kept = 0;  // inliers in the chunks before this one (same value in every thread)
    if (ok) {
        float R[9], t[3];
        for (int k = 0; k < 9; ++k) R[k] = (float)bp[k];
        for (int k = 0; k < 3; ++k) t[k] = (float)bp[9 + k];
        for (int i0 = 0; i0 < n; i0 += kBatch * nthr) {
            const PointBatch q = i0 == 0 ? first : load_batch(p, base, i0, n);
            bool in[kBatch];
            unsigned long long bal[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int i = i0 + k * nthr + tid;
                in[k] = false;
                if (i < n) {
                    float ux, uy;
                    kin.normalise(q.pu[k], q.pv[k], ux, uy);
                    float qq;  // the scoring's own expression: the mask's count IS the winner's count
                    in[k] = inlier_q(R, t, q.X[k], q.Y[k], q.Z[k], ux, uy, thr2, qq);
                    mask[i] = in[k] ? 1 : 0;
                }
                bal[k] = __ballot(in[k]);
            }
            // inlier counts of the batch's chunks per wavefront through LDS: ONE barrier pair per batch gives every thread the
            // number of inliers before its own (order-preserving compaction) and the running total (the inlier count itself)
            __syncthreads();  // cc: the caller's arg-max / the previous batch's prefix may still be reading
            if (lane < kBatch) cc[lane][wave] = __popcll(lane == 0 ? bal[0] : (lane == 1 ? bal[1] : (lane == 2 ? bal[2] : bal[3])));
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                int off = kept;
                for (int w = 0; w < nwaves; ++w) {
                    if (w < wave) off += cc[k][w];
                    kept += cc[k][w];
                }
                if (sel && in[k])
                    rows.entry_from(base, off + __popcll(bal[k] & ((1ull << lane) - 1ull)), q.pu[k], q.pv[k], q.sw[k], q.X[k], q.Y[k], q.Z[k], q.src[k]);
            }
        }
        LC_SEL_STAMP(4);
        if (tid == 0) p.n_inliers[b] = kept;
    }
    LC_SEL_STAMP(5);
    if (sel) {
        const int cnt = rows.pad(base, p.pose0 + b, n, kept, p.sel_min_count, p.sel_seed);
        if (tid == 0) p.sel_counts[b] = cnt;
    }
}

// fewer than 4 correspondences: cv2.solvePnPRansac needs >= 4 (EPnP: 5 model points); flagged invalid like a failed call
// (cv2_solver.py:74-80).  Called by every thread of the workgroup (the inlier mask is all zero: the selection keeps nothing and pads).
__device__ __forceinline__ void write_too_few(const RansacParams& p, int b, int n) {
    if (p.sel_w) {
        const int cnt = selection_rows(p).pad((size_t)b * p.Nmax, p.pose0 + b, n, 0, p.sel_min_count, p.sel_seed);
        if (threadIdx.x == 0) p.sel_counts[b] = cnt;
    }
    if (threadIdx.x != 0) return;
    float* st = p.states + 7 * (size_t)b;
    st[0] = 1; st[1] = st[2] = st[3] = st[4] = st[5] = st[6] = 0;
    p.invalid[b] = 1;
    p.n_inliers[b] = 0;
    if (p.best_hyp) p.best_hyp[b] = -1;
    if (p.valid_counts) p.valid_counts[b] = 0;
}

constexpr int kRansacMaxWaves = 4;  // hypothesis rounds of 64 run on separate wavefronts of the pose's workgroup

__global__ __launch_bounds__(64 * kRansacMaxWaves) void lc_pnp_ransac_kernel(const RansacParams p) {
    __shared__ float sx[kLdsTilePts * 3];   // 3D points of the tile
    __shared__ float su[kLdsTilePts * 2];   // normalised image coordinates K^-1 (u,v,1)
    __shared__ double best_pose[kRansacMaxWaves][12];
    __shared__ int wv_cnt[kRansacMaxWaves], wv_hyp[kRansacMaxWaves];
    __shared__ float wv_err[kRansacMaxWaves];
    __shared__ ChunkCounts chunk_cnt;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    const size_t base = (size_t)b * p.Nmax;
    unsigned char* mask = p.inlier_mask + base;
    zero_bytes(mask, 0, p.Nmax);
    if (n < 4) {
        write_too_few(p, b, max(n, 0));
        return;
    }
    const CamInv kin(p.K + 9 * (size_t)b);
    // points [t0, t0 + kLdsTilePts) of the pose -> LDS (called by the whole workgroup, barriers inside)
    int tile0 = -1;
    auto stage_tile = [&](int t0) {
        if (tile0 == t0) return;  // uniform
        if (tile0 >= 0) __syncthreads();  // the previous tile's readers
        const int cnt = min(kLdsTilePts, n - t0);
        for (int i = tid; i < cnt; i += nthr) {
            const size_t g = base + t0 + i;
            sx[3 * i] = p.pts3d[g * 3]; sx[3 * i + 1] = p.pts3d[g * 3 + 1]; sx[3 * i + 2] = p.pts3d[g * 3 + 2];
            kin.normalise(p.pts2d[g * 2], p.pts2d[g * 2 + 1], su[2 * i], su[2 * i + 1]);
        }
        tile0 = t0;
        __syncthreads();
    };
    stage_tile(0);
    // inlier threshold in normalised coordinates: reprojectionError px / focal scale (sqrt|det K2|)
    const float thr_px = threshold_px(p, b);
    const float thr = thr_px * (float)sqrt(fabs(kin.idet));
    const float thr2 = thr * thr;

    int best_cnt = -1, best_hyp = 0;
    float best_err = INFINITY;
    Pose best;
    // hypothesis id = round*64 + lane, whatever the wave count; the wavefronts walk the rounds together (the tiles of a pose with
    // more than kLdsTilePts points are staged by the whole workgroup), a wavefront without a round of its own only keeps the barriers
    for (int round0 = 0; round0 < p.rounds; round0 += nwaves) {
        const int round = round0 + wave;
        const bool active = round < p.rounds;
        Pose cand;
        bool have = false;
        if (active) {
            int idx[4];
            sample_indices(p.seed, p.pose0 + b, round * kWave + lane, n, idx);
            have = hypothesis_pose(idx, [&](int i, float (&X)[3], float (&u)[2]) {
                const int j = i - tile0;
                if (j >= 0 && j < kLdsTilePts) {  // the same floats either way: the tile holds what the other branch computes
                    X[0] = sx[3 * j]; X[1] = sx[3 * j + 1]; X[2] = sx[3 * j + 2];
                    u[0] = su[2 * j]; u[1] = su[2 * j + 1];
                } else {
                    X[0] = p.pts3d[(base + i) * 3]; X[1] = p.pts3d[(base + i) * 3 + 1]; X[2] = p.pts3d[(base + i) * 3 + 2];
                    kin.normalise(p.pts2d[(base + i) * 2], p.pts2d[(base + i) * 2 + 1], u[0], u[1]);
                }
            }, cand);
        }
        // score on ALL points (every lane walks the LDS arrays: broadcast reads)
        int cnt = have ? 0 : -1;
        float err = have ? 0.f : INFINITY;
        float R[9], t[3];
        for (int k = 0; k < 9; ++k) R[k] = have ? (float)cand.R[k] : 0.f;
        for (int k = 0; k < 3; ++k) t[k] = have ? (float)cand.t[k] : 0.f;
        for (int t0 = 0; t0 < n; t0 += kLdsTilePts) {
            stage_tile(t0);
            if (!have) continue;
            // the inlier error is added in the association of the split form (even / odd points of a 64-point chunk in two sums,
            // chunk totals in chunk order), so that both launch forms hand the SAME float to the (count, error, id) tie-break and
            // an object gets the same winner whichever form its batch size selects
            const int nt = min(kLdsTilePts, n - t0);
            for (int i0 = 0; i0 < nt; i0 += kWave) {
                float e0 = 0.f, e1 = 0.f;
                const int i1 = min(nt, i0 + kWave);
                for (int i = i0; i < i1; i += 2) {
                    score_point(R, t, sx[3 * i], sx[3 * i + 1], sx[3 * i + 2], su[2 * i], su[2 * i + 1], thr2, cnt, e0);
                    if (i + 1 < i1) score_point(R, t, sx[3 * i + 3], sx[3 * i + 4], sx[3 * i + 5], su[2 * i + 2], su[2 * i + 3], thr2, cnt, e1);
                }
                err += chunk_error(e0, e1, t[2]);
            }
        }
        if (active && (cnt > best_cnt || (cnt == best_cnt && err < best_err))) {
            best_cnt = cnt; best_err = err; best_hyp = round * kWave + lane;
            if (have) best = cand;
        }
    }
    // arg-max over lanes, then over the waves: (count, -err, -hypothesis id)
    int win_cnt = best_cnt, win_hyp = best_hyp;
    float win_err = best_err;
    auto better = better_hyp;
    for (int m = 32; m >= 1; m >>= 1) {
        const int oc = __shfl_xor(win_cnt, m, kWave), oh = __shfl_xor(win_hyp, m, kWave);
        const float oe = __shfl_xor(win_err, m, kWave);
        if (better(oc, oe, oh, win_cnt, win_err, win_hyp)) { win_cnt = oc; win_err = oe; win_hyp = oh; }
    }
    if (best_hyp == win_hyp && best_cnt == win_cnt && win_cnt >= 0) {
        for (int k = 0; k < 9; ++k) best_pose[wave][k] = best.R[k];
        for (int k = 0; k < 3; ++k) best_pose[wave][9 + k] = best.t[k];
    }
    if (lane == 0) { wv_cnt[wave] = win_cnt; wv_err[wave] = win_err; wv_hyp[wave] = win_hyp; }
    __syncthreads();
    int ww = 0;
    win_cnt = wv_cnt[0]; win_err = wv_err[0]; win_hyp = wv_hyp[0];
    for (int w = 1; w < nwaves; ++w) {
        if (better(wv_cnt[w], wv_err[w], wv_hyp[w], win_cnt, win_err, win_hyp)) { win_cnt = wv_cnt[w]; win_err = wv_err[w]; win_hyp = wv_hyp[w]; ww = w; }
    }
    write_result(p, b, n, win_cnt >= 4, best_pose[ww], win_hyp, thr2, kin, chunk_cnt, load_batch(p, base, 0, n));
}


// ---- split form (lc_pnp_ransac_init3_f32): the same RANSAC as three launches over a caller-provided workspace ---------------------
// The one-workgroup-per-pose kernel above keeps one pose on ONE compute unit: 64 objects x 150 hypotheses x 1000+ points use a
// quarter of the chip, and every wavefront walks all points of its pose (1024 x ~22 VALU slots x 4 cycles = 38 us at best).  Here
//   1. hypotheses: grid rounds x B, one lane per hypothesis: sample, P3P, fourth-point pick -> workspace (fp32 for scoring, fp64
//      for the answer);
//   2. scoring: one wavefront per (pose, chunk of 64 points, round of 64 hypotheses): the chunk is staged in LDS, every lane scores
//      its hypothesis on it and writes (count, error) of the chunk -- no atomics: the chunk partials are summed in chunk order by
//      step 3, so results do not depend on scheduling;
//   3. selection: grid B: arg-max of (count, -error, -hypothesis id), inlier mask of the winner over all points, outputs (and the
//      inlier re-selection).  Bound by its load instructions: the pose's count first, then one round trip sized by it (select_winner).
// Same hypothesis stream, same per-point arithmetic, same ordering as the single launch (the error sums are associated by chunk).
constexpr int kChunkPts = 64;

__host__ __device__ inline unsigned long long pack_partial(int cnt, float err) {
    return (unsigned long long)(unsigned)cnt | ((unsigned long long)__builtin_bit_cast(unsigned, err) << 32);
}
__host__ __device__ inline int partial_cnt(unsigned long long v) { return (int)(unsigned)v; }
__host__ __device__ inline float partial_err(unsigned long long v) { return __builtin_bit_cast(float, (unsigned)(v >> 32)); }

__host__ __device__ inline size_t ransac_counter_bytes(int B) { return 16 * (((size_t)B + 3) / 4); }  // keeps the double2 loads behind it 16-byte aligned

struct RansacWorkspace {
    unsigned* arrived;  // (B,) chunks of the pose scored so far (ticketed form; zeroed by the hypotheses launch)
    double* hyp64;   // (B, H, 12)
    float* hyp32;    // (B, H, 12)
    unsigned long long* part;  // (B, C, H) chunk partials: inlier count in the low word, the bits of the float error sum in the high word
    int H, C;
};
__host__ __device__ inline RansacWorkspace carve_workspace(void* ws, int B, int Nmax, int rounds) {
    RansacWorkspace w;
    w.H = rounds * kWave;
    w.C = (Nmax + kChunkPts - 1) / kChunkPts;  // every point of a row is scored
    if (w.C < 1) w.C = 1;
    char* q = static_cast<char*>(ws);
    w.arrived = reinterpret_cast<unsigned*>(q); q += ransac_counter_bytes(B);
    w.hyp64 = reinterpret_cast<double*>(q); q += sizeof(double) * 12 * (size_t)B * w.H;
    w.hyp32 = reinterpret_cast<float*>(q); q += sizeof(float) * 12 * (size_t)B * w.H;
    w.part = reinterpret_cast<unsigned long long*>(q);
    return w;
}

__global__ __launch_bounds__(kWave) void lc_ransac_hypotheses_kernel(const RansacParams p) {
    LC_P3P_STAMP(0);
    const int b = blockIdx.x / p.rounds, hyp = (blockIdx.x % p.rounds) * kWave + threadIdx.x;
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    if (hyp == 0) w.arrived[b] = 0;  // the ticketed scoring launch that follows counts the pose's chunks from zero
    if (n < 4) return;  // the selection step flags the pose
    const CamInv kin(p.K + 9 * (size_t)b);
    const size_t base = (size_t)b * p.Nmax;
    int idx[4];
    sample_indices(p.seed, p.pose0 + b, hyp, n, idx);
    LC_P3P_STAMP(1);
    Pose cand;
    const bool have = hypothesis_pose(idx, [&](int i, float (&X)[3], float (&u)[2]) {
        X[0] = p.pts3d[(base + i) * 3]; X[1] = p.pts3d[(base + i) * 3 + 1]; X[2] = p.pts3d[(base + i) * 3 + 2];
        kin.normalise(p.pts2d[(base + i) * 2], p.pts2d[(base + i) * 2 + 1], u[0], u[1]);
    }, cand);
    LC_P3P_STAMP(5);
    double* o64 = w.hyp64 + 12 * ((size_t)b * w.H + hyp);
    float* o32 = w.hyp32 + 12 * ((size_t)b * w.H + hyp);
    if (!have) {  // behind the camera for every point: no inliers (the single launch scores such a lane -1, below every usable count)
#pragma unroll
        for (int k = 0; k < 9; ++k) cand.R[k] = 0.0;
        cand.t[0] = cand.t[1] = 0.0; cand.t[2] = -1.0;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { o64[k] = cand.R[k]; o32[k] = (float)cand.R[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { o64[9 + k] = cand.t[k]; o32[9 + k] = (float)cand.t[k]; }
    LC_P3P_STAMP(6);
}

__global__ __launch_bounds__(kWave * kRansacMaxWaves) void lc_ransac_score_kernel(const RansacParams p) {
    // One wavefront per (pose, chunk, round of 64 hypotheses); the four wavefronts of a workgroup are independent (own LDS slice, no
    // barrier).  Structure of arrays: four consecutive points load as one 16-byte LDS read per coordinate, already paired for the
    // packed math.
    __shared__ __attribute__((aligned(16))) float lds[kRansacMaxWaves][5][kChunkPts];
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long unit = (long long)blockIdx.x * kRansacMaxWaves + wave;
    const int per_pose = w.C * p.rounds;
    if (unit >= (long long)p.B * per_pose) return;
    const int b = (int)(unit / per_pose), rem = (int)(unit % per_pose), c = rem / p.rounds, round = rem % p.rounds;  // neighbours share a chunk
    float *sX = lds[wave][0], *sY = lds[wave][1], *sZ = lds[wave][2], *sU = lds[wave][3], *sV = lds[wave][4];
    // Everything that does not depend on the pose's point count is requested first (the count itself, K, this lane's hypothesis,
    // the chunk's points up to the padded row length): one memory round trip instead of three dependent ones.
    const int i0 = c * kChunkPts, cap = max(0, min(kChunkPts, p.Nmax - i0));
    const size_t base = (size_t)b * p.Nmax + i0;
    static_assert(kChunkPts == kWave, "one point per lane");
    const bool in_row = lane < cap;
    const float gx = in_row ? p.pts3d[(base + lane) * 3] : 0.f, gy = in_row ? p.pts3d[(base + lane) * 3 + 1] : 0.f,
                gz = in_row ? p.pts3d[(base + lane) * 3 + 2] : 0.f;
    const float gu = in_row ? p.pts2d[(base + lane) * 2] : 0.f, gv = in_row ? p.pts2d[(base + lane) * 2 + 1] : 0.f;
    const int hyp = round * kWave + lane;
    float R[9], t[3];
    {
        const float* h32 = w.hyp32 + 12 * ((size_t)b * w.H + hyp);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = h32[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = h32[9 + k];
    }
    const CamInv kin(p.K + 9 * (size_t)b);
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    if (n < 4) return;
    const int cnt_pts = max(0, min(kChunkPts, n - i0)), cnt4 = (cnt_pts + 3) & ~3;
    if (cnt_pts == 0) return;  // the selection step sums the chunks the pose has
    if (lane < cnt_pts) {
        float ux, uy;
        kin.normalise(gu, gv, ux, uy);
        sU[lane] = -ux; sV[lane] = -uy;  // stored negated: the residual is one fma
        sX[lane] = gx; sY[lane] = gy; sZ[lane] = gz;
    } else if (lane < cnt4) {  // padding of the last group of four: an infinite image coordinate is never an inlier
        sX[lane] = sY[lane] = sZ[lane] = 0.f;
        sU[lane] = sV[lane] = -INFINITY;
    }
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wavefront execute in order: no wait beyond the compiler's own
    const float thr_px = threshold_px(p, b);
    const float thr = thr_px * (float)sqrt(fabs(kin.idet));
    const float thr2 = thr * thr;
    int cnt = 0;
    v2f_t err2 = {0.f, 0.f};
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto rd = [](const float* a, int i) { return *reinterpret_cast<const v4f_t*>(a + i); };
    // two groups of four points per iteration, in two register sets: the LDS reads of one are in flight while the other is scored
    const int last = __builtin_amdgcn_readfirstlane(cnt4) - 4;
    v4f_t X = rd(sX, 0), Y = rd(sY, 0), Z = rd(sZ, 0), U = rd(sU, 0), V = rd(sV, 0);
    for (int i = 0; i <= last; i += 8) {
        const int i1 = min(i + 4, last), i2 = min(i + 8, last);
        const v4f_t X1 = rd(sX, i1), Y1 = rd(sY, i1), Z1 = rd(sZ, i1), U1 = rd(sU, i1), V1 = rd(sV, i1);
        score_pair(R, t, X.xy, Y.xy, Z.xy, U.xy, V.xy, thr2, cnt, err2);
        score_pair(R, t, X.zw, Y.zw, Z.zw, U.zw, V.zw, thr2, cnt, err2);
        X = rd(sX, i2); Y = rd(sY, i2); Z = rd(sZ, i2); U = rd(sU, i2); V = rd(sV, i2);
        if (i + 4 <= last) {  // uniform
            score_pair(R, t, X1.xy, Y1.xy, Z1.xy, U1.xy, V1.xy, thr2, cnt, err2);
            score_pair(R, t, X1.zw, Y1.zw, Z1.zw, U1.zw, V1.zw, thr2, cnt, err2);
        }
    }
    const size_t o = ((size_t)b * w.C + c) * w.H + hyp;
    w.part[o] = pack_partial(cnt, chunk_error(err2.x, err2.y, t[2]));
}

// Rows wider than 4096 candidates (zlmo's test-time shape: 16 384 per object, of which a pose's count keeps a fifth): one wavefront per
// (pose, GROUP of kWideGroup consecutive chunks, round of 64 hypotheses).  The pose's count is read FIRST -- most groups of such a row lie
// behind it and leave at once (the one-chunk kernel above requests its points before it knows: 49 k wavefronts, four fifths of them for
// nothing, 42 us) -- the hypothesis is loaded once per group, and the next chunk's points are in flight while this one is scored.  Same
// per-point arithmetic, one partial per 64-point chunk as before: the selection step sees the same numbers.
#ifndef LC_WIDE_GROUP
#define LC_WIDE_GROUP 1  // (-D: A/B of the group size.  zlmo, 64 x ~3300 of 16 384 points, 3 rounds: 8 -> 62.4, 4 -> 42.2, 2 -> 40.3, 1 -> 37.7 us: the
                         // kernel wants MORE wavefronts per SIMD to hide its chains behind, not fewer hypothesis loads; profiles/r04/NOTES.md)
#endif
constexpr int kWideGroup = LC_WIDE_GROUP;
__global__ __launch_bounds__(kWave * kRansacMaxWaves) void lc_ransac_score_wide_kernel(const RansacParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kRansacMaxWaves][5][kChunkPts];
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long unit = (long long)blockIdx.x * kRansacMaxWaves + wave;
    const int groups = (w.C + kWideGroup - 1) / kWideGroup, per_pose = groups * p.rounds;
    if (unit >= (long long)p.B * per_pose) return;
    const int b = (int)(unit / per_pose), rem = (int)(unit % per_pose), c0 = (rem / p.rounds) * kWideGroup, round = rem % p.rounds;
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    if (n < 4 || c0 * kChunkPts >= n) return;
    const int c1 = min(min(c0 + kWideGroup, w.C), (n + kChunkPts - 1) / kChunkPts);  // chunks [c0, c1) of this pose exist
    float *sX = lds[wave][0], *sY = lds[wave][1], *sZ = lds[wave][2], *sU = lds[wave][3], *sV = lds[wave][4];
    const size_t row = (size_t)b * p.Nmax;
    struct Pt { float x, y, z, u, v; };
    auto fetch = [&](int c) {  // lane's point of chunk c (rows are padded to Nmax: every index below Nmax may be read)
        const size_t i = row + min(c * kChunkPts + lane, p.Nmax - 1);
        return Pt{p.pts3d[i * 3], p.pts3d[i * 3 + 1], p.pts3d[i * 3 + 2], p.pts2d[i * 2], p.pts2d[i * 2 + 1]};
    };
    Pt cur = fetch(c0);
    const int hyp = round * kWave + lane;
    float R[9], t[3];
    {
        const float* h32 = w.hyp32 + 12 * ((size_t)b * w.H + hyp);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = h32[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = h32[9 + k];
    }
    const CamInv kin(p.K + 9 * (size_t)b);
    const float thr_px = threshold_px(p, b);
    const float thr = thr_px * (float)sqrt(fabs(kin.idet));
    const float thr2 = thr * thr;
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto rd = [](const float* a, int i) { return *reinterpret_cast<const v4f_t*>(a + i); };
    for (int c = c0; c < c1; ++c) {
        const Pt nxt = fetch(c + 1 < c1 ? c + 1 : c);  // branch-free: the last iteration re-requests its own chunk (a cache hit)
        const int i0 = c * kChunkPts, cnt_pts = min(kChunkPts, n - i0), cnt4 = (cnt_pts + 3) & ~3;
        if (lane < cnt_pts) {
            float ux, uy;
            kin.normalise(cur.u, cur.v, ux, uy);
            sU[lane] = -ux; sV[lane] = -uy;
            sX[lane] = cur.x; sY[lane] = cur.y; sZ[lane] = cur.z;
        } else if (lane < cnt4) {
            sX[lane] = sY[lane] = sZ[lane] = 0.f;
            sU[lane] = sV[lane] = -INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;
        v2f_t err2 = {0.f, 0.f};
        // two groups of four points per iteration, in two register sets: the LDS reads of one are in flight while the other is scored (with one
        // set the compiler rotates the loop and every iteration waits out its own reads).  cnt4 is wave-uniform: a scalar loop.
        const int last = __builtin_amdgcn_readfirstlane(cnt4) - 4;
        v4f_t X = rd(sX, 0), Y = rd(sY, 0), Z = rd(sZ, 0), U = rd(sU, 0), V = rd(sV, 0);
        for (int i = 0; i <= last; i += 8) {
            const int i1 = min(i + 4, last), i2 = min(i + 8, last);
            const v4f_t X1 = rd(sX, i1), Y1 = rd(sY, i1), Z1 = rd(sZ, i1), U1 = rd(sU, i1), V1 = rd(sV, i1);
            score_pair(R, t, X.xy, Y.xy, Z.xy, U.xy, V.xy, thr2, cnt, err2);
            score_pair(R, t, X.zw, Y.zw, Z.zw, U.zw, V.zw, thr2, cnt, err2);
            X = rd(sX, i2); Y = rd(sY, i2); Z = rd(sZ, i2); U = rd(sU, i2); V = rd(sV, i2);
            if (i + 4 <= last) {  // uniform
                score_pair(R, t, X1.xy, Y1.xy, Z1.xy, U1.xy, V1.xy, thr2, cnt, err2);
                score_pair(R, t, X1.zw, Y1.zw, Z1.zw, U1.zw, V1.zw, thr2, cnt, err2);
            }
        }
        w.part[((size_t)b * w.C + c) * w.H + hyp] = pack_partial(cnt, chunk_error(err2.x, err2.y, t[2]));
        __builtin_amdgcn_wave_barrier();  // the chunk's LDS reads precede the next chunk's writes (one wavefront: program order)
        cur = nxt;
    }
}

// The same for at most 128 poses (the test-time batches of the dense heads), without the wavefronts that leave at once: at zlmo's shape four fifths of
// the grid above -- 39 k of 49 k wavefronts, 9.8 k of 12.3 k workgroups -- read a count and exit, and the kernel's 37 us are what DISPATCHING 12 k
// workgroups takes (the SQ counters of profiles/r04/test_time/pmc_zlmo.md: 1.7 wavefronts resident per SIMD on average, VALU-active 29 % of the wave
// cycles).  Here a fixed grid of five wavefronts per SIMD walks the LIVE (pose, chunk, round) units: every wavefront forms the prefix of the poses'
// unit counts (lane l: poses l and l + 64) and takes units w, w + W, ...; the unit's pose falls out of two ballots.  Same arithmetic per unit, same
// partials.
#ifndef LC_LIVE_WORKGROUPS
#define LC_LIVE_WORKGROUPS 2048  // (768: 24.2, 1024: 23.2, 1280: 23.1, 2048: 22.5, 2560: 22.7 us at zlmo's shape)
#endif
constexpr int kLiveMaxPoses = 2 * kWave, kLiveWorkgroups = LC_LIVE_WORKGROUPS;
__global__ __launch_bounds__(kWave * kRansacMaxWaves) void lc_ransac_score_live_kernel(const RansacParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kRansacMaxWaves][5][kChunkPts];
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = (int)blockIdx.x * kRansacMaxWaves + wave, stride = (int)gridDim.x * kRansacMaxWaves;
    auto count_of = [&](int b) { return b < p.B ? min(p.counts ? p.counts[b] : p.Nmax, p.Nmax) : 0; };
    const int n_lo = count_of(lane), n_hi = count_of(lane + kWave);
    auto units_of = [&](int n) { return n >= 4 ? ((n + kChunkPts - 1) / kChunkPts) * p.rounds : 0; };
    int incl_lo = units_of(n_lo), incl_hi = units_of(n_hi);  // -> inclusive prefixes in pose order (poses 0..63, then 64..127)
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int a = __shfl_up(incl_lo, d, kWave), c = __shfl_up(incl_hi, d, kWave);
        if (lane >= d) { incl_lo += a; incl_hi += c; }
    }
    incl_hi += __shfl(incl_lo, kWave - 1, kWave);
    const int live = __shfl(incl_hi, kWave - 1, kWave);
    float *sX = lds[wave][0], *sY = lds[wave][1], *sZ = lds[wave][2], *sU = lds[wave][3], *sV = lds[wave][4];
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto rd = [](const float* a, int i) { return *reinterpret_cast<const v4f_t*>(a + i); };
    for (int u = first; u < live; u += stride) {  // wave-uniform
        // the pose whose units hold u: the first whose inclusive prefix exceeds it (poses without units share their predecessor's prefix)
        const int b = __builtin_amdgcn_readfirstlane(__popcll(__ballot(incl_lo <= u)) + __popcll(__ballot(incl_hi <= u)));
        const int before = b == 0 ? 0 : (b <= kWave ? __shfl(incl_lo, b - 1, kWave) : __shfl(incl_hi, b - 1 - kWave, kWave));
        const int n = b < kWave ? __shfl(n_lo, b, kWave) : __shfl(n_hi, b - kWave, kWave);
        const int rem = __builtin_amdgcn_readfirstlane(u - before), c = rem / p.rounds, round = rem - c * p.rounds;  // neighbours share a chunk
        const size_t i = (size_t)b * p.Nmax + min(c * kChunkPts + lane, p.Nmax - 1);
        const float gx = p.pts3d[i * 3], gy = p.pts3d[i * 3 + 1], gz = p.pts3d[i * 3 + 2], gu = p.pts2d[i * 2], gv = p.pts2d[i * 2 + 1];
        const int hyp = round * kWave + lane;
        float R[9], t[3];
        {
            const float* h32 = w.hyp32 + 12 * ((size_t)b * w.H + hyp);
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = h32[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = h32[9 + k];
        }
        const CamInv kin(p.K + 9 * (size_t)b);
        const float thr_px = threshold_px(p, b);
        const float thr = thr_px * (float)sqrt(fabs(kin.idet));
        const float thr2 = thr * thr;
        const int i0 = c * kChunkPts, cnt_pts = min(kChunkPts, n - i0), cnt4 = (cnt_pts + 3) & ~3;
        if (lane < cnt_pts) {
            float ux, uy;
            kin.normalise(gu, gv, ux, uy);
            sU[lane] = -ux; sV[lane] = -uy;
            sX[lane] = gx; sY[lane] = gy; sZ[lane] = gz;
        } else if (lane < cnt4) {
            sX[lane] = sY[lane] = sZ[lane] = 0.f;
            sU[lane] = sV[lane] = -INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;
        v2f_t err2 = {0.f, 0.f};
        const int last = __builtin_amdgcn_readfirstlane(cnt4) - 4;
        v4f_t X = rd(sX, 0), Y = rd(sY, 0), Z = rd(sZ, 0), U = rd(sU, 0), V = rd(sV, 0);
        for (int j = 0; j <= last; j += 8) {
            const int j1 = min(j + 4, last), j2 = min(j + 8, last);
            const v4f_t X1 = rd(sX, j1), Y1 = rd(sY, j1), Z1 = rd(sZ, j1), U1 = rd(sU, j1), V1 = rd(sV, j1);
            score_pair(R, t, X.xy, Y.xy, Z.xy, U.xy, V.xy, thr2, cnt, err2);
            score_pair(R, t, X.zw, Y.zw, Z.zw, U.zw, V.zw, thr2, cnt, err2);
            X = rd(sX, j2); Y = rd(sY, j2); Z = rd(sZ, j2); U = rd(sU, j2); V = rd(sV, j2);
            if (j + 4 <= last) {  // uniform
                score_pair(R, t, X1.xy, Y1.xy, Z1.xy, U1.xy, V1.xy, thr2, cnt, err2);
                score_pair(R, t, X1.zw, Y1.zw, Z1.zw, U1.zw, V1.zw, thr2, cnt, err2);
            }
        }
        w.part[((size_t)b * w.C + c) * w.H + hyp] = pack_partial(cnt, chunk_error(err2.x, err2.y, t[2]));
        __builtin_amdgcn_wave_barrier();  // the unit's LDS reads precede the next unit's writes (one wavefront: program order)
    }
}

// Selection of pose b by the calling workgroup: chunk partials of every hypothesis summed in chunk order (the sums do not depend on
// which workgroup finished first), arg-max of (count, -error, -hypothesis id), outputs.  XCD: the partials were written by other
// workgroups of THIS launch (read around the caches), else by an earlier launch.
constexpr int kMoreChunks = 16;  // chunk partials a selection thread requests per round trip behind the first kFirstChunks (template argument)
struct SelectShared {
    double best_pose[12];
    int wv_cnt[kSelMaxWaves], wv_hyp[kSelMaxWaves];
    float wv_err[kSelMaxWaves];
    ChunkCounts chunk_cnt;
};
template <bool XCD, int kFirstChunks = 32>
__device__ __forceinline__ void select_winner(const RansacParams& p, const RansacWorkspace& w, int b, SelectShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    auto part_at = [&](size_t o) { return XCD ? xcd_load(w.part + o) : w.part[o]; };
    int win_cnt = -1, win_hyp = 0x7fffffff;
    float win_err = INFINITY;
    LC_SEL_STAMP(0);
    // The pose's point count first (one scalar), then everything that does not depend on the winner in ONE memory round trip, and no
    // more of it than the count asks for (the kernel is bound by the number of its load instructions: 75 per thread with all 16
    // chunk rows and all four batch slots requested blindly, ~30 this way): the first batch of the pose's correspondences (the
    // inlier mask below needs them), this thread's own hypothesis in double precision (the winner's is the answer: its thread hands
    // it over through LDS instead of a dependent load), the chunk partials of this thread's first hypothesis.
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    const int chunks = n >= 4 ? (n + kChunkPts - 1) / kChunkPts : 0;
    double2 mine[6];
    {
        const double2* h = reinterpret_cast<const double2*>(w.hyp64 + 12 * ((size_t)b * w.H + (tid < w.H ? tid : 0)));
#pragma unroll
        for (int k = 0; k < 6; ++k) mine[k] = h[k];
    }
    unsigned long long pc0[kFirstChunks];
#pragma unroll
    for (int c = 0; c < kFirstChunks; ++c)
        pc0[c] = (c < chunks && tid < w.H) ? part_at(((size_t)b * w.C + c) * w.H + tid) : 0ull;
    // (the batch last: its values pass through selects, i.e. a wait -- requested first, the two groups above were issued behind that wait)
    const PointBatch first = load_batch(p, (size_t)b * p.Nmax, 0, n);
    const CamInv kin(p.K + 9 * (size_t)b);
    unsigned char* mask = p.inlier_mask + (size_t)b * p.Nmax;
    zero_bytes(mask, 0, p.Nmax);
    if (n < 4) {
        write_too_few(p, b, max(n, 0));
        return;
    }
    if (tid < w.H) {
        int cnt = 0;
        float err = 0.f;
#pragma unroll
        for (int c = 0; c < kFirstChunks; ++c) {  // chunk order; rows the pose does not have were not requested (zero)
            cnt += partial_cnt(pc0[c]);
            err += c < chunks ? partial_err(pc0[c]) : 0.f;
        }
        // a pose with more than kFirstChunks * 64 points: the further chunks, still in chunk order, kMoreChunks requests in flight
        for (int c0 = kFirstChunks; c0 < chunks; c0 += kMoreChunks) {
            unsigned long long pc[kMoreChunks];
#pragma unroll
            for (int k = 0; k < kMoreChunks; ++k) pc[k] = c0 + k < chunks ? part_at(((size_t)b * w.C + c0 + k) * w.H + tid) : 0ull;
#pragma unroll
            for (int k = 0; k < kMoreChunks; ++k) {
                cnt += partial_cnt(pc[k]);
                err += c0 + k < chunks ? partial_err(pc[k]) : 0.f;
            }
        }
        win_cnt = cnt; win_err = err; win_hyp = tid;
    }
    LC_SEL_STAMP(1);
    for (int hyp = tid + nthr; hyp < w.H; hyp += nthr) {  // more hypotheses than threads
        int cnt = 0;
        float err = 0.f;
        for (int c = 0; c < chunks; ++c) {
            const unsigned long long v = part_at(((size_t)b * w.C + c) * w.H + hyp);
            cnt += partial_cnt(v);
            err += partial_err(v);
        }
        if (better_hyp(cnt, err, hyp, win_cnt, win_err, win_hyp)) { win_cnt = cnt; win_err = err; win_hyp = hyp; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const int oc = __shfl_xor(win_cnt, m, kWave), oh = __shfl_xor(win_hyp, m, kWave);
        const float oe = __shfl_xor(win_err, m, kWave);
        if (better_hyp(oc, oe, oh, win_cnt, win_err, win_hyp)) { win_cnt = oc; win_err = oe; win_hyp = oh; }
    }
    if (lane == 0) { sh.wv_cnt[wave] = win_cnt; sh.wv_err[wave] = win_err; sh.wv_hyp[wave] = win_hyp; }
    __syncthreads();
    win_cnt = sh.wv_cnt[0]; win_err = sh.wv_err[0]; win_hyp = sh.wv_hyp[0];
    for (int v = 1; v < nwaves; ++v)
        if (better_hyp(sh.wv_cnt[v], sh.wv_err[v], sh.wv_hyp[v], win_cnt, win_err, win_hyp)) { win_cnt = sh.wv_cnt[v]; win_err = sh.wv_err[v]; win_hyp = sh.wv_hyp[v]; }
    const bool ok = win_cnt >= 4;
    LC_SEL_STAMP(2);
    if (ok) {
        if (win_hyp < nthr) {
            if (tid == win_hyp)
#pragma unroll
                for (int k = 0; k < 6; ++k) { sh.best_pose[2 * k] = mine[k].x; sh.best_pose[2 * k + 1] = mine[k].y; }
        } else if (tid < 12) {  // more hypotheses than threads and the winner is one of the later ones
            sh.best_pose[tid] = w.hyp64[12 * ((size_t)b * w.H + win_hyp) + tid];
        }
    }
    __syncthreads();
    const float thr_px = threshold_px(p, b);
    const float thr = thr_px * (float)sqrt(fabs(kin.idet));
    LC_SEL_STAMP(3);
    write_result(p, b, n, ok, sh.best_pose, win_hyp, thr * thr, kin, sh.chunk_cnt, first);
    LC_SEL_STAMP(6);
}

__global__ __launch_bounds__(kWave * kRansacMaxWaves) void lc_ransac_select_kernel(const RansacParams p) {
    __shared__ SelectShared sh;
    select_winner<false>(p, carve_workspace(p.workspace, p.B, p.Nmax, p.rounds), blockIdx.x, sh);
}
// rows of more than 4096 candidates: 16 wavefronts write the inlier mask and the re-selection (4096 points per pass and barrier pair instead
// of 1024); fewer chunk partials are requested blindly (128 VGPRs per thread at this workgroup size)
__global__ __launch_bounds__(kWave * kSelMaxWaves) void lc_ransac_select_wide_kernel(const RansacParams p) {
    __shared__ SelectShared sh;
    select_winner<false, 8>(p, carve_workspace(p.workspace, p.B, p.Nmax, p.rounds), blockIdx.x, sh);
}

// Ticketed form (lc_pnp_ransac_init4_f32 with `ticketed`): scoring AND selection in one launch.  One workgroup per (pose, chunk of 64 points), its
// wavefronts take the rounds of 64 hypotheses; the chunk's partials are written through the caches, the workgroup counts itself
// in, and the workgroup that completes the pose's count runs the selection -- nobody waits, so no co-residency is assumed, and the
// partials are still summed in chunk order: same results as the three launches.  Saves the selection launch and its boundary --
// and MEASURED SLOWER than the three launches on MI355X (64 objects x 1024 points, replayed: 29.8 vs 26.9 us,
// profiles/r03/test_time/ransac_forms.txt): write-through stores, their acknowledgement, the arrival atomic and the partials read around the
// caches cost more than the 1.45 us kernel boundary they replace.  Kept as an option (tests compare it with the three launches);
// the default is the three launches.
__global__ __launch_bounds__(kWave * kRansacMaxWaves) void lc_ransac_score_select_kernel(const RansacParams p) {
    __shared__ __attribute__((aligned(16))) float lds[5][kChunkPts];
    __shared__ SelectShared sh;
    __shared__ unsigned s_before;
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x / w.C, c = blockIdx.x % w.C;
    float *sX = lds[0], *sY = lds[1], *sZ = lds[2], *sU = lds[3], *sV = lds[4];
    // requested before the pose's point count is known, as in the scoring kernel above
    const int i0 = c * kChunkPts, cap = max(0, min(kChunkPts, p.Nmax - i0));
    const size_t base = (size_t)b * p.Nmax + i0;
    const bool in_row = wave == 0 && lane < cap;
    const float gx = in_row ? p.pts3d[(base + lane) * 3] : 0.f, gy = in_row ? p.pts3d[(base + lane) * 3 + 1] : 0.f,
                gz = in_row ? p.pts3d[(base + lane) * 3 + 2] : 0.f;
    const float gu = in_row ? p.pts2d[(base + lane) * 2] : 0.f, gv = in_row ? p.pts2d[(base + lane) * 2 + 1] : 0.f;
    float R[9], t[3];
    {
        const float* h32 = w.hyp32 + 12 * ((size_t)b * w.H + min(wave, p.rounds - 1) * kWave + lane);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = h32[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = h32[9 + k];
    }
    const CamInv kin(p.K + 9 * (size_t)b);
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    if (n < 4) {  // nothing was scored: chunk 0's workgroup flags the pose
        if (c == 0) {
            unsigned char* mask = p.inlier_mask + (size_t)b * p.Nmax;
            zero_bytes(mask, 0, p.Nmax);
            write_too_few(p, b, max(n, 0));
        }
        return;
    }
    const int chunks = (n + kChunkPts - 1) / kChunkPts;
    const int cnt_pts = max(0, min(kChunkPts, n - i0)), cnt4 = (cnt_pts + 3) & ~3;
    if (cnt_pts == 0) return;  // a chunk the pose does not have: not counted
    if (wave == 0) {
        if (lane < cnt_pts) {
            float ux, uy;
            kin.normalise(gu, gv, ux, uy);
            sU[lane] = -ux; sV[lane] = -uy;
            sX[lane] = gx; sY[lane] = gy; sZ[lane] = gz;
        } else if (lane < cnt4) {
            sX[lane] = sY[lane] = sZ[lane] = 0.f;
            sU[lane] = sV[lane] = -INFINITY;
        }
    }
    __syncthreads();
    const float thr_px = threshold_px(p, b);
    const float thr = thr_px * (float)sqrt(fabs(kin.idet));
    const float thr2 = thr * thr;
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto rd = [](const float* a, int i) { return *reinterpret_cast<const v4f_t*>(a + i); };
    for (int round = wave; round < p.rounds; round += nwaves) {
        const int hyp = round * kWave + lane;
        if (round != wave) {
            const float* h32 = w.hyp32 + 12 * ((size_t)b * w.H + hyp);
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = h32[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = h32[9 + k];
        }
        int cnt = 0;
        v2f_t err2 = {0.f, 0.f};
        v4f_t X = rd(sX, 0), Y = rd(sY, 0), Z = rd(sZ, 0), U = rd(sU, 0), V = rd(sV, 0);
        for (int i = 0; i < cnt4; i += 4) {
            const int nx = i + 4 < cnt4 ? i + 4 : i;
            const v4f_t Xn = rd(sX, nx), Yn = rd(sY, nx), Zn = rd(sZ, nx), Un = rd(sU, nx), Vn = rd(sV, nx);
            score_pair(R, t, X.xy, Y.xy, Z.xy, U.xy, V.xy, thr2, cnt, err2);
            score_pair(R, t, X.zw, Y.zw, Z.zw, U.zw, V.zw, thr2, cnt, err2);
            X = Xn; Y = Yn; Z = Zn; U = Un; V = Vn;
        }
        const size_t o = ((size_t)b * w.C + c) * w.H + hyp;
        xcd_store(w.part + o, pack_partial(cnt, chunk_error(err2.x, err2.y, t[2])));
    }
    xcd_stores_done();
    __syncthreads();  // every wave's partials have been acknowledged
    if (tid == 0) s_before = __hip_atomic_fetch_add(w.arrived + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_before + 1u != (unsigned)chunks) return;
    select_winner<true>(p, w, b, sh);
}

}  // namespace

size_t pnp_ransac_workspace_bytes(int B, int Nmax, int rounds) {
    if (B <= 0 || rounds <= 0) return 0;
    const int H = rounds * kWave;
    const int C = Nmax > 0 ? (Nmax + kChunkPts - 1) / kChunkPts : 1;
    return ransac_counter_bytes(B) + (size_t)B * H * 12 * (sizeof(double) + sizeof(float)) + (size_t)B * C * H * (sizeof(int) + sizeof(float));
}

// where carve_workspace puts things, for the diagnostics that read the workspace off the device (include/lc_amd.h: lc_pnp_ransac_workspace_layout)
void pnp_ransac_workspace_layout(int B, int Nmax, int rounds, size_t out[6]) {
    const RansacWorkspace w = carve_workspace(nullptr, B, Nmax, rounds);
    out[0] = reinterpret_cast<size_t>(w.hyp64);
    out[1] = reinterpret_cast<size_t>(w.hyp32);
    out[2] = reinterpret_cast<size_t>(w.part);
    out[3] = (size_t)w.H;
    out[4] = (size_t)w.C;
    out[5] = pnp_ransac_workspace_bytes(B, Nmax, rounds);
}

int launch_pnp_ransac(const RansacParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    const int waves = p.rounds < kRansacMaxWaves ? (p.rounds < 1 ? 1 : p.rounds) : kRansacMaxWaves;
    if (!p.workspace) {  // single launch, one workgroup per pose
        hipLaunchKernelGGL(lc_pnp_ransac_kernel, dim3(p.B), dim3(64 * waves), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (p.workspace_bytes < pnp_ransac_workspace_bytes(p.B, p.Nmax, p.rounds)) return 3;
    const RansacWorkspace w = carve_workspace(p.workspace, p.B, p.Nmax, p.rounds);
    hipLaunchKernelGGL(lc_ransac_hypotheses_kernel, dim3((unsigned)p.rounds * p.B), dim3(kWave), 0, stream, p);
    const long long units = (long long)p.B * w.C * p.rounds;
    if (p.ticketed) {
        hipLaunchKernelGGL(lc_ransac_score_select_kernel, dim3((unsigned)p.B * w.C), dim3(kWave * waves), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
#ifndef LC_LIVE_MIN_CHUNKS
#define LC_LIVE_MIN_CHUNKS 64  // (-D: A/B of the live-unit kernel on narrower rows)
#endif
    if (w.C > LC_LIVE_MIN_CHUNKS) {  // more than 4096 candidates per row: chunk groups, the count first
        const long long wide = (long long)p.B * ((w.C + kWideGroup - 1) / kWideGroup) * p.rounds;
        if (p.B <= kLiveMaxPoses)
            hipLaunchKernelGGL(lc_ransac_score_live_kernel, dim3((unsigned)std::min<long long>(kLiveWorkgroups, (wide + kRansacMaxWaves - 1) / kRansacMaxWaves)),
                               dim3(kWave * kRansacMaxWaves), 0, stream, p);
        else
            hipLaunchKernelGGL(lc_ransac_score_wide_kernel, dim3((unsigned)((wide + kRansacMaxWaves - 1) / kRansacMaxWaves)), dim3(kWave * kRansacMaxWaves), 0, stream, p);
    } else {
        hipLaunchKernelGGL(lc_ransac_score_kernel, dim3((unsigned)((units + kRansacMaxWaves - 1) / kRansacMaxWaves)), dim3(kWave * kRansacMaxWaves), 0, stream, p);
    }
    if (w.C > 64) hipLaunchKernelGGL(lc_ransac_select_wide_kernel, dim3(p.B), dim3(kWave * kSelMaxWaves), 0, stream, p);
    else hipLaunchKernelGGL(lc_ransac_select_kernel, dim3(p.B), dim3(kWave * kRansacMaxWaves), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc

#ifdef LC_P3P_STAMPS
extern "C" __attribute__((visibility("default"))) int lc_debug_sel_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::p3p_diag::g_sel_stamp), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int lc_debug_p3p_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::p3p_diag::g_p3p_stamp), sizeof(unsigned long long) * 7) == hipSuccess ? 0 : 1;
}
#endif
