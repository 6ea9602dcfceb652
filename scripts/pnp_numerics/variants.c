/* Numerics experiment (NOT product, NOT oracle): re-includes the oracle's trust-region loop with other linear solvers
 * for the LM step, to measure how often the integer output `rets` differs from the DENSE_QR restatement.
 *   kind 0  the oracle's Householder QR (sanity: must give zero flips)
 *   kind 1  the same QR with every column sum accumulated in REVERSE row order (a last-bit perturbation of the same
 *           algorithm: the flip rate two correct DENSE_QR implementations have against each other)
 *   kind 2  normal equations H = A^T A, LDL^T in double (what lc_pnp_body.h did in round 1)
 *   kind 3  kind 2 + one step of iterative refinement with the residual formed from A itself (corrected semi-normal eq.)
 *   kind 4  normal equations accumulated in double-double (error-free products + compensated sums), LDL^T in double-double
 */
#include <math.h>
static int variant_solver(double *A, double *b, int m, double y[6]);
#define PNP_STEP_SOLVER variant_solver
#include "../../oracle/pnp_lm_oracle.c"

static int g_kind = 0;
void variants_set_kind(int k) { g_kind = k; }

static int qr_rev(double *A, double *b, int m, double y[6]) {
    for (int j = 0; j < 6; ++j) {
        double nrm = 0;
        for (int i = m - 1; i >= j; --i) nrm += A[i * 6 + j] * A[i * 6 + j];
        nrm = sqrt(nrm);
        if (!(nrm > 0) || !isfinite(nrm)) return 0;
        const double alpha = A[j * 6 + j] > 0 ? -nrm : nrm;
        const double v0 = A[j * 6 + j] - alpha;
        double vnorm2 = 0;
        for (int i = m - 1; i > j; --i) vnorm2 += A[i * 6 + j] * A[i * 6 + j];
        vnorm2 += v0 * v0;
        if (vnorm2 > 0) {
            for (int c = j + 1; c < 6; ++c) {
                double d = 0;
                for (int i = m - 1; i > j; --i) d += A[i * 6 + j] * A[i * 6 + c];
                d += v0 * A[j * 6 + c];
                d = 2 * d / vnorm2;
                A[j * 6 + c] -= d * v0;
                for (int i = j + 1; i < m; ++i) A[i * 6 + c] -= d * A[i * 6 + j];
            }
            double d = 0;
            for (int i = m - 1; i > j; --i) d += A[i * 6 + j] * b[i];
            d += v0 * b[j];
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

static int ldlt6(const double H[6][6], const double g[6], double y[6]) {
    double L[6][6], d[6];
    for (int j = 0; j < 6; ++j) {
        double dj = H[j][j];
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
        if (!(dj > 0) || !isfinite(dj)) return 0;
        d[j] = dj;
        for (int i = j + 1; i < 6; ++i) {
            double v = H[j][i];
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v / dj;
        }
    }
    double z[6];
    for (int i = 0; i < 6; ++i) { double v = g[i]; for (int k = 0; k < i; ++k) v -= L[i][k] * z[k]; z[i] = v; }
    for (int i = 5; i >= 0; --i) { double v = z[i] / d[i]; for (int k = i + 1; k < 6; ++k) v -= L[k][i] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; ++i) if (!isfinite(y[i])) return 0;
    return 1;
}

static int normal_eq(double *A, double *b, int m, double y[6], int refine) {
    double H[6][6], g[6];
    for (int i = 0; i < 6; ++i) {
        for (int j = i; j < 6; ++j) { double s = 0; for (int r = 0; r < m; ++r) s += A[r * 6 + i] * A[r * 6 + j]; H[i][j] = H[j][i] = s; }
        double s = 0; for (int r = 0; r < m; ++r) s += A[r * 6 + i] * b[r]; g[i] = s;
    }
    if (!ldlt6(H, g, y)) return 0;
    for (int it = 0; it < refine; ++it) {
        double g2[6] = {0, 0, 0, 0, 0, 0}, dy[6];
        for (int r = 0; r < m; ++r) {
            double e = b[r];
            for (int j = 0; j < 6; ++j) e -= A[r * 6 + j] * y[j];
            for (int j = 0; j < 6; ++j) g2[j] += A[r * 6 + j] * e;
        }
        if (!ldlt6(H, g2, dy)) return 0;
        for (int j = 0; j < 6; ++j) y[j] += dy[j];
    }
    return 1;
}

/* double-double helpers */
typedef struct { double h, l; } dd;
static inline dd two_sum(double a, double b) { double s = a + b, bb = s - a; dd r = {s, (a - (s - bb)) + (b - bb)}; return r; }
static inline dd two_prod(double a, double b) { double p = a * b; dd r = {p, fma(a, b, -p)}; return r; }
static inline dd dd_add(dd a, dd b) { dd s = two_sum(a.h, b.h); s.l += a.l + b.l; return two_sum(s.h, s.l); }
static inline dd dd_neg(dd a) { dd r = {-a.h, -a.l}; return r; }
static inline dd dd_mul(dd a, dd b) { dd p = two_prod(a.h, b.h); p.l += a.h * b.l + a.l * b.h; return two_sum(p.h, p.l); }
static inline dd dd_div(dd a, dd b) {
    double q1 = a.h / b.h;
    dd t = dd_mul((dd){q1, 0}, b);
    dd r = dd_add(a, dd_neg(t));
    double q2 = r.h / b.h;
    return two_sum(q1, q2);
}
static int normal_eq_dd(double *A, double *b, int m, double y[6]) {
    dd H[6][6], g[6];
    for (int i = 0; i < 6; ++i) {
        for (int j = i; j < 6; ++j) { dd s = {0, 0}; for (int r = 0; r < m; ++r) s = dd_add(s, two_prod(A[r * 6 + i], A[r * 6 + j])); H[i][j] = H[j][i] = s; }
        dd s = {0, 0}; for (int r = 0; r < m; ++r) s = dd_add(s, two_prod(A[r * 6 + i], b[r])); g[i] = s;
    }
    dd L[6][6], d[6];
    for (int j = 0; j < 6; ++j) {
        dd dj = H[j][j];
        for (int k = 0; k < j; ++k) dj = dd_add(dj, dd_neg(dd_mul(dd_mul(L[j][k], L[j][k]), d[k])));
        if (!(dj.h > 0) || !isfinite(dj.h)) return 0;
        d[j] = dj;
        for (int i = j + 1; i < 6; ++i) {
            dd v = H[j][i];
            for (int k = 0; k < j; ++k) v = dd_add(v, dd_neg(dd_mul(dd_mul(L[i][k], L[j][k]), d[k])));
            L[i][j] = dd_div(v, dj);
        }
    }
    dd z[6], yy[6];
    for (int i = 0; i < 6; ++i) { dd v = g[i]; for (int k = 0; k < i; ++k) v = dd_add(v, dd_neg(dd_mul(L[i][k], z[k]))); z[i] = v; }
    for (int i = 5; i >= 0; --i) { dd v = dd_div(z[i], d[i]); for (int k = i + 1; k < 6; ++k) v = dd_add(v, dd_neg(dd_mul(L[k][i], yy[k]))); yy[i] = v; }
    for (int i = 0; i < 6; ++i) { y[i] = yy[i].h + yy[i].l; if (!isfinite(y[i])) return 0; }
    return 1;
}

static int variant_solver(double *A, double *b, int m, double y[6]) {
    switch (g_kind) {
        case 0: return qr_solve6(A, b, m, y);
        case 1: return qr_rev(A, b, m, y);
        case 2: return normal_eq(A, b, m, y, 0);
        case 3: return normal_eq(A, b, m, y, 1);
        default: return normal_eq_dd(A, b, m, y);
    }
}
