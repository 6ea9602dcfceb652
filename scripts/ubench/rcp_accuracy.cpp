// Diagnostic: relative error of v_rcp_f64 / v_rsq_f64 raw and after Newton steps (decides how many steps fast_rcp needs).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y0 = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, y0, 1.0);
    double y1 = __builtin_fma(y0, e, y0);
    e = __builtin_fma(-v, y1, 1.0);
    double y2 = __builtin_fma(y1, e, y1);
    double r0 = __builtin_amdgcn_rsq(v);
    double h = 0.5 * r0, g = v * r0;            // one Newton-Raphson (Goldschmidt) step for sqrt/rsqrt
    double rr = __builtin_fma(-h, g, 0.5);
    double g1 = __builtin_fma(g, rr, g), h1 = __builtin_fma(h, rr, h);
    double rr2 = __builtin_fma(-h1, g1, 0.5);
    double g2 = __builtin_fma(g1, rr2, g1);
    out[6 * i] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y2; out[6 * i + 3] = r0; out[6 * i + 4] = g1; out[6 * i + 5] = g2;
}
int main() {
    const int n = 1 << 16;
    double *hx = new double[n], *ho = new double[6 * n], *dx, *dout;
    for (int i = 0; i < n; ++i) hx[i] = std::exp((i - n / 2) * 1e-3) * (1 + 0.37 * std::sin(i));
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double m[6] = {0};
    for (int i = 0; i < n; ++i) {
        double x = std::fabs(hx[i]);
        double ref[6] = {1 / hx[i], 1 / hx[i], 1 / hx[i], 1 / std::sqrt(x), std::sqrt(x), std::sqrt(x)};
        double in = hx[i] > 0 ? 1 : 0;
        for (int j = 0; j < 6; ++j) {
            if (j >= 3 && !in) continue;
            double e = std::fabs(ho[6 * i + j] - ref[j]) / std::fabs(ref[j]);
            if (e > m[j]) m[j] = e;
        }
    }
    printf("rcp raw %.3e, +1 Newton %.3e, +2 Newton %.3e | rsq raw %.3e, sqrt +1 step %.3e, +2 steps %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
