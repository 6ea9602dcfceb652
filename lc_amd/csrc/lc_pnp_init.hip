// GPU initialiser for the weighted PnP solve: RANSAC over minimal P3P hypotheses.
//
// Takes the place of lib/pnp/cv2_solver.py:69-88 (cv2.solvePnPRansac, EPnP kernel, 150 iterations, multiprocessing.Pool)
// in front of cer_solver.solve (test.py:59,120): one wavefront per pose, one hypothesis per lane and round (64 x rounds
// hypotheses >= the reference's 150), every lane scores its own hypothesis against all correspondences staged in LDS,
// the wave keeps the hypothesis with most inliers (ties: smaller inlier error), and writes the pose + the inlier mask.
// No device->host->device trip, no process pool.  OpenCV's RNG/EPnP cannot be reproduced bit for bit (and OpenCV is absent
// here: parity at this boundary is unpinned and outside the metric); the contract kept is the role: a pose
// inside the LM basin of convergence plus an inlier set for `weighted_filtered` (test.py:129-134).
//
// P3P: depths l_i of three bearings y_i with |l_i y_i - l_j y_j|^2 = |x_i - x_j|^2.  The pencil D1 + g D2 of the two
// constant-free quadrics is made singular by a root g of a cubic (coefficients from 3x3 determinants), the singular quadric
// splits into two planes through its eigen-decomposition, each plane cuts D1 in <= 2 rays (a quadratic), the scale comes
// from one distance constraint; depths are polished by Gauss-Newton and R,t follow from the three point pairs.
#include <cfloat>

#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

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
    return (1.0 / sqrt(fmax(n, DBL_MIN))) * v;
}

struct Pose { double R[9], t[3]; };

// up to 4 poses; returns their count
__device__ int p3p(const V3 (&y)[3], const V3 (&x)[3], Pose (&out)[4]) {
    const double c12 = dot(y[0], y[1]), c13 = dot(y[0], y[2]), c23 = dot(y[1], y[2]);
    const V3 d12 = x[0] - x[1], d13 = x[0] - x[2], d23 = x[1] - x[2];
    const double a12 = dot(d12, d12), a13 = dot(d13, d13), a23 = dot(d23, d23);
    const V3 dn = cross(d12, d13);
    if (!(dot(dn, dn) > 1e-12 * a12 * a13)) return 0;  // collinear sample
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
    // real roots of the cubic (trigonometric / Cardano on the depressed form), Newton-polished
    double roots[3];
    int nroots = 0;
    if (fabs(k3) > 1e-14 * (fabs(k0) + fabs(k1) + fabs(k2) + fabs(k3))) {
        const double b = k2 / k3, c = k1 / k3, d = k0 / k3;
        const double p = c - b * b / 3.0, q = 2.0 * b * b * b / 27.0 - b * c / 3.0 + d;
        const double disc = q * q / 4.0 + p * p * p / 27.0;
        if (disc > 0) {
            const double s = sqrt(disc);
            roots[nroots++] = cbrt(-q / 2.0 + s) + cbrt(-q / 2.0 - s) - b / 3.0;
        } else {
            const double m = 2.0 * sqrt(fmax(-p / 3.0, 0.0));
            const double arg = m > 0 ? fmin(1.0, fmax(-1.0, 3.0 * q / (p * m))) : 0.0;
            const double th = acos(arg) / 3.0;
            for (int k = 0; k < 3; ++k) roots[nroots++] = m * cos(th - 2.0943951023931953 * k) - b / 3.0;
        }
        for (int r = 0; r < nroots; ++r) {
            double g = roots[r];
            for (int it = 0; it < 3; ++it) {
                const double f = ((g + b) * g + c) * g + d, fp = (3.0 * g + 2.0 * b) * g + c;
                if (fabs(fp) > 0) g -= f / fp;
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
    // X^-1 for the rotation recovery: X = [d12, d13, d12 x d13] (columns)
    double Xi[9];
    {
        const double X[9] = {d12.x, d13.x, dn.x, d12.y, d13.y, dn.y, d12.z, d13.z, dn.z};
        const double dt = det3(X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], X[8]);
        const double id = 1.0 / dt;
        Xi[0] = (X[4] * X[8] - X[5] * X[7]) * id; Xi[1] = (X[2] * X[7] - X[1] * X[8]) * id; Xi[2] = (X[1] * X[5] - X[2] * X[4]) * id;
        Xi[3] = (X[5] * X[6] - X[3] * X[8]) * id; Xi[4] = (X[0] * X[8] - X[2] * X[6]) * id; Xi[5] = (X[2] * X[3] - X[0] * X[5]) * id;
        Xi[6] = (X[3] * X[7] - X[4] * X[6]) * id; Xi[7] = (X[1] * X[6] - X[0] * X[7]) * id; Xi[8] = (X[0] * X[4] - X[1] * X[3]) * id;
    }
    int nsol = 0;
    for (int r = 0; r < nroots && nsol == 0; ++r) {
        const double g = roots[r];
        const Sym3 D0{D1.a00 + g * D2.a00, D1.a01 + g * D2.a01, D1.a02 + g * D2.a02, D1.a11 + g * D2.a11, D1.a12 + g * D2.a12,
                      D1.a22 + g * D2.a22};
        // eigenvalues s1, s2 of the rank-2 matrix: s1 + s2 = trace, s1 s2 = sum of principal 2x2 minors
        const double tr = D0.a00 + D0.a11 + D0.a22;
        const double m2 = D0.a00 * D0.a11 - D0.a01 * D0.a01 + D0.a00 * D0.a22 - D0.a02 * D0.a02 + D0.a11 * D0.a22 - D0.a12 * D0.a12;
        if (!(m2 < 0)) continue;  // needs eigenvalues of opposite sign to split into two real planes
        const double sq = sqrt(tr * tr - 4.0 * m2);
        const double s1 = 0.5 * (tr + sq), s2 = 0.5 * (tr - sq);  // s1 > 0 > s2
        const Sym3 S1{D0.a00 - s1, D0.a01, D0.a02, D0.a11 - s1, D0.a12, D0.a22 - s1};
        const V3 e1 = null_vec(S1), e0 = null_vec(D0);
        const V3 e2 = cross(e0, e1);
        const double sgm = sqrt(-s2 / s1);
        for (int sign = 0; sign < 2; ++sign) {
            const V3 pl = e1 + ((sign ? -sgm : sgm) * e2);  // plane pl . lambda = 0
            // eliminate the component with the largest |pl|: lambda_k = u lambda_i + v lambda_j
            const double ax = fabs(pl.x), ay = fabs(pl.y), az = fabs(pl.z);
            const int k = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
            const int i = k == 0 ? 1 : 0, j = k == 2 ? 1 : 2;
            const double pk = k == 0 ? pl.x : (k == 1 ? pl.y : pl.z), pi = i == 0 ? pl.x : pl.y, pj = j == 1 ? pl.y : pl.z;
            const double u = -pi / pk, v = -pj / pk;
            auto ent = [&](int r_, int c_) {
                const int lo = r_ < c_ ? r_ : c_, hi = r_ < c_ ? c_ : r_;
                return lo == 0 ? (hi == 0 ? D1.a00 : (hi == 1 ? D1.a01 : D1.a02)) : (lo == 1 ? (hi == 1 ? D1.a11 : D1.a12) : D1.a22);
            };
            const double Dii = ent(i, i), Djj = ent(j, j), Dkk = ent(k, k), Dij = ent(i, j), Dik = ent(i, k), Djk = ent(j, k);
            const double A = Dii + Dkk * u * u + 2.0 * Dik * u;
            const double C = Djj + Dkk * v * v + 2.0 * Djk * v;
            const double Bq = 2.0 * (Dkk * u * v + Dij + Dik * v + Djk * u);
            // A + Bq tau + C tau^2 = 0, tau = lambda_j / lambda_i
            double taus[2];
            int nt = 0;
            if (fabs(C) > 1e-14 * (fabs(A) + fabs(Bq) + fabs(C))) {
                const double disc = Bq * Bq - 4.0 * A * C;
                if (disc >= 0) {
                    const double s = sqrt(disc), qq = -0.5 * (Bq + (Bq >= 0 ? s : -s));
                    taus[nt++] = qq / C;
                    if (qq != 0) taus[nt++] = A / qq;
                }
            } else if (fabs(Bq) > 0) {
                taus[nt++] = -A / Bq;
            }
            const double cij = (i == 0 && j == 1) ? c12 : ((i == 0 && j == 2) ? c13 : c23);
            const double aij = (i == 0 && j == 1) ? a12 : ((i == 0 && j == 2) ? a13 : a23);
            for (int tt = 0; tt < nt && nsol < 4; ++tt) {
                const double tau = taus[tt];
                if (!(tau > 0)) continue;
                const double den = 1.0 + tau * tau - 2.0 * cij * tau;
                if (!(den > 0)) continue;
                const double li = sqrt(aij / den), lj = tau * li, lk = u * li + v * lj;
                if (!(lk > 0)) continue;
                double l[3];
                l[i] = li; l[j] = lj; l[k] = lk;
                // Gauss-Newton polish of the three distance constraints
                for (int it = 0; it < 3; ++it) {
                    const double r0 = l[0] * l[0] + l[1] * l[1] - 2.0 * c12 * l[0] * l[1] - a12;
                    const double r1 = l[0] * l[0] + l[2] * l[2] - 2.0 * c13 * l[0] * l[2] - a13;
                    const double r2 = l[1] * l[1] + l[2] * l[2] - 2.0 * c23 * l[1] * l[2] - a23;
                    const double J00 = 2.0 * (l[0] - c12 * l[1]), J01 = 2.0 * (l[1] - c12 * l[0]);
                    const double J10 = 2.0 * (l[0] - c13 * l[2]), J12 = 2.0 * (l[2] - c13 * l[0]);
                    const double J21 = 2.0 * (l[1] - c23 * l[2]), J22 = 2.0 * (l[2] - c23 * l[1]);
                    const double dt = det3(J00, J01, 0, J10, 0, J12, 0, J21, J22);
                    if (!(fabs(dt) > 1e-300)) break;
                    const double id = 1.0 / dt;
                    l[0] -= det3(r0, J01, 0, r1, 0, J12, r2, J21, J22) * id;
                    l[1] -= det3(J00, r0, 0, J10, r1, J12, 0, r2, J22) * id;
                    l[2] -= det3(J00, J01, r0, J10, 0, r1, 0, J21, r2) * id;
                }
                if (!(l[0] > 0 && l[1] > 0 && l[2] > 0)) continue;
                const V3 z1 = l[0] * y[0], z2 = l[1] * y[1], z3 = l[2] * y[2];
                const V3 yd1 = z1 - z2, yd2 = z1 - z3, yn = cross(yd1, yd2);
                const double Y[9] = {yd1.x, yd2.x, yn.x, yd1.y, yd2.y, yn.y, yd1.z, yd2.z, yn.z};
                Pose& o = out[nsol];
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) o.R[3 * a + c] = Y[3 * a] * Xi[c] + Y[3 * a + 1] * Xi[3 + c] + Y[3 * a + 2] * Xi[6 + c];
                const V3 Rx = {o.R[0] * x[0].x + o.R[1] * x[0].y + o.R[2] * x[0].z, o.R[3] * x[0].x + o.R[4] * x[0].y + o.R[5] * x[0].z,
                               o.R[6] * x[0].x + o.R[7] * x[0].y + o.R[8] * x[0].z};
                o.t[0] = z1.x - Rx.x; o.t[1] = z1.y - Rx.y; o.t[2] = z1.z - Rx.z;
                ++nsol;
            }
        }
    }
    return nsol;
}

__device__ __forceinline__ unsigned hash_u32(unsigned a) {  // lowbias32
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}

// rotation matrix -> quaternion (w,x,y,z), w >= 0
__device__ void mat_to_quat(const double R[9], float q[4]) {
    const double tr = R[0] + R[4] + R[8];
    double w, x, y, z;
    if (tr > 0) {
        const double s = sqrt(tr + 1.0) * 2.0;
        w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
        w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
        w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
    } else {
        const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
        w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
    }
    const double n = 1.0 / sqrt(w * w + x * x + y * y + z * z), sg = w < 0 ? -n : n;
    q[0] = (float)(w * sg); q[1] = (float)(x * sg); q[2] = (float)(y * sg); q[3] = (float)(z * sg);
}

constexpr int kMaxLdsPts = 2048;

constexpr int kRansacMaxWaves = 4;  // hypothesis rounds of 64 run on separate wavefronts of the pose's workgroup

__global__ __launch_bounds__(64 * kRansacMaxWaves) void lc_pnp_ransac_kernel(const RansacParams p) {
    __shared__ float sx[kMaxLdsPts * 3];   // 3D points
    __shared__ float su[kMaxLdsPts * 2];   // normalised image coordinates K^-1 (u,v,1)
    __shared__ double best_pose[kRansacMaxWaves][12];
    __shared__ int wv_cnt[kRansacMaxWaves], wv_hyp[kRansacMaxWaves];
    __shared__ float wv_err[kRansacMaxWaves];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    const int n = min(p.counts ? p.counts[b] : p.Nmax, p.Nmax);
    const size_t base = (size_t)b * p.Nmax;
    unsigned char* mask = p.inlier_mask + base;
    for (int i = tid; i < p.Nmax; i += nthr) mask[i] = 0;
    if (n < 4) {  // cv2.solvePnPRansac needs >= 4 (EPnP: 5 model points); flagged invalid like a failed call (cv2_solver.py:74-80)
        if (tid == 0) {
            float* st = p.states + 7 * (size_t)b;
            st[0] = 1; st[1] = st[2] = st[3] = st[4] = st[5] = st[6] = 0;
            p.invalid[b] = 1;
            p.n_inliers[b] = 0;
            if (p.best_hyp) p.best_hyp[b] = -1;
        }
        return;
    }
    const float* Kp = p.K + 9 * (size_t)b;
    // u = k0 X/Z + k1 Y/Z + k2, v = k3 X/Z + k4 Y/Z + k5  (row 2 of K is (0,0,1) as in ceres.cpp:46-47)
    const double k0 = Kp[0], k1 = Kp[1], k2 = Kp[2], k3 = Kp[3], k4 = Kp[4], k5 = Kp[5];
    const double idet = 1.0 / (k0 * k4 - k1 * k3);
    const int nl = min(n, kMaxLdsPts);  // hypotheses are scored on the first kMaxLdsPts points (dense heads: N <= 1849)
    for (int i = tid; i < nl; i += nthr) {
        sx[3 * i] = p.pts3d[(base + i) * 3]; sx[3 * i + 1] = p.pts3d[(base + i) * 3 + 1]; sx[3 * i + 2] = p.pts3d[(base + i) * 3 + 2];
        const double du = (double)p.pts2d[(base + i) * 2] - k2, dv = (double)p.pts2d[(base + i) * 2 + 1] - k5;
        su[2 * i] = (float)((k4 * du - k1 * dv) * idet);
        su[2 * i + 1] = (float)((-k3 * du + k0 * dv) * idet);
    }
    __syncthreads();
    // inlier threshold in normalised coordinates: reprojectionError px / focal scale (sqrt|det K2|)
    const float thr_px = p.reproj_err_per_pose ? p.reproj_err_per_pose[b] : p.reproj_err;
    const float thr = thr_px * (float)sqrt(fabs(idet));
    const float thr2 = thr * thr;

    int best_cnt = -1, best_hyp = 0;
    float best_err = INFINITY;
    Pose best;
    for (int round = wave; round < p.rounds; round += nwaves) {  // hypothesis id = round*64 + lane, whatever the wave count
        // minimal sample: 3 distinct indices + a 4th to disambiguate the up-to-4 P3P solutions
        unsigned h = hash_u32(p.seed ^ hash_u32((unsigned)b * 0x9E3779B9u + (unsigned)(round * kWave + lane)));
        int idx[4];
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
        V3 y[3], x[3];
        for (int k = 0; k < 3; ++k) {
            const double ux = su[2 * idx[k]], uy = su[2 * idx[k] + 1];
            const double inv = 1.0 / sqrt(ux * ux + uy * uy + 1.0);
            y[k] = {ux * inv, uy * inv, inv};
            x[k] = {sx[3 * idx[k]], sx[3 * idx[k] + 1], sx[3 * idx[k] + 2]};
        }
        Pose sols[4];
        const int ns = p3p(y, x, sols);
        // pick the solution that reprojects the 4th point best
        int pick = -1;
        float pick_e = INFINITY;
        for (int s = 0; s < ns; ++s) {
            const float X = sx[3 * idx[3]], Y = sx[3 * idx[3] + 1], Z = sx[3 * idx[3] + 2];
            const float cx = (float)sols[s].R[0] * X + (float)sols[s].R[1] * Y + (float)sols[s].R[2] * Z + (float)sols[s].t[0];
            const float cy = (float)sols[s].R[3] * X + (float)sols[s].R[4] * Y + (float)sols[s].R[5] * Z + (float)sols[s].t[1];
            const float cz = (float)sols[s].R[6] * X + (float)sols[s].R[7] * Y + (float)sols[s].R[8] * Z + (float)sols[s].t[2];
            if (!(cz > 0)) continue;
            const float ex = cx / cz - su[2 * idx[3]], ey = cy / cz - su[2 * idx[3] + 1];
            const float e = ex * ex + ey * ey;
            if (e < pick_e) { pick_e = e; pick = s; }
        }
        // score on all points (every lane walks the LDS arrays: broadcast reads)
        int cnt = -1;
        float err = INFINITY;
        if (pick >= 0) {
            float R[9], t[3];
            for (int k = 0; k < 9; ++k) R[k] = (float)sols[pick].R[k];
            for (int k = 0; k < 3; ++k) t[k] = (float)sols[pick].t[k];
            cnt = 0;
            err = 0.f;
            for (int i = 0; i < nl; ++i) {
                const float X = sx[3 * i], Y = sx[3 * i + 1], Z = sx[3 * i + 2];
                const float cz = R[6] * X + R[7] * Y + R[8] * Z + t[2];
                const float icz = __builtin_amdgcn_rcpf(cz);  // 1 ulp: scoring only
                const float ex = (R[0] * X + R[1] * Y + R[2] * Z + t[0]) * icz - su[2 * i];
                const float ey = (R[3] * X + R[4] * Y + R[5] * Z + t[1]) * icz - su[2 * i + 1];
                const float e = ex * ex + ey * ey;
                const bool in = cz > 0 && e < thr2;
                cnt += in ? 1 : 0;
                err += in ? e : 0.f;
            }
        }
        if (cnt > best_cnt || (cnt == best_cnt && err < best_err)) {
            best_cnt = cnt; best_err = err; best_hyp = round * kWave + lane;
            if (pick >= 0) best = sols[pick];
        }
    }
    // arg-max over lanes, then over the waves: (count, -err, -hypothesis id)
    int win_cnt = best_cnt, win_hyp = best_hyp;
    float win_err = best_err;
    auto better = [](int oc, float oe, int oh, int c, float e, int h) { return oc > c || (oc == c && (oe < e || (oe == e && oh < h))); };
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
    const double* bp = best_pose[ww];
    const bool ok = win_cnt >= 4;
    if (ok) {  // inlier mask of the winner over ALL n points
        float R[9], t[3];
        for (int k = 0; k < 9; ++k) R[k] = (float)bp[k];
        for (int k = 0; k < 3; ++k) t[k] = (float)bp[9 + k];
        int total = 0;
        for (int i = tid; i < n; i += nthr) {
            const float X = p.pts3d[(base + i) * 3], Y = p.pts3d[(base + i) * 3 + 1], Z = p.pts3d[(base + i) * 3 + 2];
            const double du = (double)p.pts2d[(base + i) * 2] - k2, dv = (double)p.pts2d[(base + i) * 2 + 1] - k5;
            const float ux = (float)((k4 * du - k1 * dv) * idet), uy = (float)((-k3 * du + k0 * dv) * idet);
            const float cz = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            const float ex = (R[0] * X + R[1] * Y + R[2] * Z + t[0]) / cz - ux, ey = (R[3] * X + R[4] * Y + R[5] * Z + t[1]) / cz - uy;
            const bool in = cz > 0 && (ex * ex + ey * ey) < thr2;
            mask[i] = in ? 1 : 0;
            total += in ? 1 : 0;
        }
        for (int m = 32; m >= 1; m >>= 1) total += __shfl_xor(total, m, kWave);
        __syncthreads();  // wv_cnt was read by every thread above
        if (lane == 0) wv_cnt[wave] = total;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < nwaves; ++w) tot += wv_cnt[w];
            p.n_inliers[b] = tot;
        }
    }
    if (tid == 0) {
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
    }
}

}  // namespace

int launch_pnp_ransac(const RansacParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    const int waves = p.rounds < kRansacMaxWaves ? (p.rounds < 1 ? 1 : p.rounds) : kRansacMaxWaves;
    hipLaunchKernelGGL(lc_pnp_ransac_kernel, dim3(p.B), dim3(64 * waves), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
