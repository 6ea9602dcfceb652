// Host check of lc_common.h: atan_ratio_pos (same arithmetic, libm sqrt and exact reciprocals standing in for the in-kernel 1-ulp
// forms): max relative error against atan2 over random unit-quaternion (s, c) pairs.   g++ -O2 atan_accuracy.cpp && ./a.out
#include <cmath>
#include <cstdio>
#include <random>
static double atan_ratio_pos(double s, double c) {
    const bool swap = s > c;
    const double a = swap ? c : s, b = swap ? s : c;
    double t = a / b;
    for (int k = 0; k < 2; ++k) t = t / (1.0 + std::sqrt(std::fma(t, t, 1.0)));
    const double w = t * t;
    double p = -1.0 / 21.0;
    const double cs[] = {1.0 / 19, -1.0 / 17, 1.0 / 15, -1.0 / 13, 1.0 / 11, -1.0 / 9, 1.0 / 7, -1.0 / 5, 1.0 / 3};
    for (double q : cs) p = std::fma(p, w, q);
    const double at = 4.0 * std::fma(-(p * w), t, t);
    return swap ? 1.57079632679489661923 - at : at;
}
int main() {
    std::mt19937_64 g(1);
    std::normal_distribution<double> n(0, 1);
    double worst = 0, worst_abs = 0;
    for (int i = 0; i < 2000000; ++i) {
        double q[4] = {n(g), n(g), n(g), n(g)};
        if (i % 4 == 0) q[0] *= 1e-6;   // near pi/2
        if (i % 4 == 1) { q[1] *= 1e-7; q[2] *= 1e-7; q[3] *= 1e-7; }  // tiny rotations
        const double s = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), c = std::fabs(q[0]);
        if (s == 0 && c == 0) continue;
        const double ref = std::atan2(s, c), got = atan_ratio_pos(s, c);
        worst_abs = std::fmax(worst_abs, std::fabs(got - ref));
        if (ref > 0) worst = std::fmax(worst, std::fabs(got - ref) / ref);
    }
    std::printf("max relative error %.3g, max absolute error %.3g\n", worst, worst_abs);
    return worst < 1e-15 ? 0 : 1;
}
