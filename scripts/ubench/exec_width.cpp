// Micro-benchmark (diagnostic, not product): does a fp64 VALU instruction cost fewer issue cycles when only part of the wave is active?
// (Wave-uniform algebra of the LM solve runs redundantly in 64 lanes: if a 16-lane EXEC issued in one pass instead of four, that algebra
// could run on a quarter wave.)  One wave per CU, s_memtime around unrolled chains under EXEC masks of 64 / 32 / 16 / 1 lanes.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/exec_width.cpp -o scripts/ubench/exec_width && scripts/ubench/exec_width
#include <hip/hip_runtime.h>
#include <cstdio>

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double seed, int width) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d = a + 1, e = a + 2, f = a + 3;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < width) {  // EXEC = the first `width` lanes for the whole chain
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 256; ++i) {
            if (MODE == 0) { a = __builtin_fma(a, b, c); }
            if (MODE == 1) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c); }
            if (MODE == 2) { float x = (float)a; x = __builtin_fmaf(x, 1.0000001f, 0.5f); a = x; }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t1);
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + d + e + f;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int width) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 0, 0, out, cyc, 1.5, width);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long best = ~0ull;
    for (auto v : h) best = v < best ? v : best;
    printf("%-28s EXEC = %2d lanes: %6.2f cycles per iteration\n", name, width, best / 256.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {64, 32, 16, 1}) run<0>("dependent v_fma_f64", w);
    for (int w : {64, 32, 16, 1}) run<1>("4 independent v_fma_f64", w);
    for (int w : {64, 32, 16, 1}) run<2>("cvt + f32 fma + cvt", w);
    return 0;
}
