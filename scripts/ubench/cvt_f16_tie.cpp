// f32 -> f16 conversion of a tie in the fp16 subnormal range: v_cvt_f16_f32 against gfx950's packed v_cvt_pk_f16_f32 (what the compiler picks for
// two conversions side by side) against round-to-nearest-even done in integers.   hipcc --offload-arch=gfx950 -O3 cvt_f16_tie.cpp -o cvt_f16_tie
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* in, unsigned short* single, unsigned short* packed, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    unsigned r1, r2;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r1) : "v"(v));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r2) : "v"(v), "v"(v));
    single[i] = (unsigned short)(r1 & 0xffff);
    packed[i] = (unsigned short)(r2 & 0xffff);
}
static unsigned short rne(float f) {  // exact RNE via double arithmetic on the fp16 grid
    double a = fabs((double)f);
    unsigned short s = f < 0 ? 0x8000 : 0;
    if (a < 6.103515625e-05) { double q = a / 5.9604644775390625e-08; double r = nearbyint(q); return s | (unsigned short)r; }
    int e; double m = frexp(a, &e);  // a = m 2^e, m in [0.5,1)
    double q = ldexp(m, 11); double r = nearbyint(q); if (r == 2048) { r = 1024; e++; }
    return s | (unsigned short)(((e + 14) << 10) + ((int)r - 1024));
}
int main() {
    const int n = 1 << 20;
    float* h = new float[n];
    unsigned seed = 12345;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const int k = (seed >> 8) % 2047;          // subnormal / low normal grid index
        const int which = (seed >> 20) % 3;        // tie, just below, just above
        double v = (k + 0.5) * 5.9604644775390625e-08;
        float f = (float)v;
        if (which == 1) f = nextafterf(f, 0.f);
        if (which == 2) f = nextafterf(f, 1.f);
        h[i] = (seed & 1) ? -f : f;
    }
    h[0] = -6.109476089477539e-06f;
    float* d; unsigned short *a, *b;
    hipMalloc(&d, n * 4); hipMalloc(&a, n * 2); hipMalloc(&b, n * 2);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, n);
    unsigned short* ha = new unsigned short[n]; unsigned short* hb = new unsigned short[n];
    hipMemcpy(ha, a, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 2, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) { unsigned short r = rne(h[i]); bad1 += ha[i] != r; bad2 += hb[i] != r; }
    printf("first: value %.10g single %04x packed %04x rne %04x\n", h[0], ha[0], hb[0], rne(h[0]));
    printf("of %d values around ties: v_cvt_f16_f32 differs from RNE on %d, v_cvt_pk_f16_f32 on %d\n", n, bad1, bad2);
    return 0;
}
