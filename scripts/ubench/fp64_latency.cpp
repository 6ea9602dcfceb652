// Micro-benchmark (diagnostic, not product): per-wave issue interval and dependent latency of the fp64 VALU / cross-lane
// instructions the LM and loss kernels are made of.  One wave per CU, s_memtime around unrolled chains.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/fp64_latency.cpp -o /tmp/fp64_latency && /tmp/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d = a + 1, e = a + 2, f = a + 3;
    unsigned long long t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        if (MODE == 0) { a = __builtin_fma(a, b, c); }                                                            // dependent fma
        if (MODE == 1) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c); }  // 4 independent
        if (MODE == 2) { a = a * b; }                                                                             // dependent mul
        if (MODE == 3) { a = a + c; }                                                                             // dependent add
        if (MODE == 4) { a = __builtin_amdgcn_rcp(a) + c; }                                                       // rcp + add
        if (MODE == 5) {                                                                                          // permlane32 swap pair + add
            auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(d), false, false);
            auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(d), false, false);
            a = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
        }
        if (MODE == 6) {                                                                                          // dpp mov pair + add
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x140, 0xF, 0xF, false);
            int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x140, 0xF, 0xF, false);
            a = a + __hiloint2double(hi, lo);
        }
        if (MODE == 7) { float x = (float)a; x = __builtin_fmaf(x, 1.0000001f, 0.5f); a = x; }                    // cvt + f32 fma + cvt
        if (MODE == 8) { a = sqrt(a); }
        if (MODE == 9) { a = 1.0 / a + c; }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    out[blockIdx.x * 64 + threadIdx.x] = a + d + e + f;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ops) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long best = ~0ull;
    for (auto v : h) best = v < best ? v : best;
    printf("%-34s %6.1f cycles per iteration (%d instr)\n", name, best / 256.0, ops);
}

int main() {
    run<0>("dependent v_fma_f64", 1);
    run<1>("4 independent v_fma_f64", 4);
    run<2>("dependent v_mul_f64", 1);
    run<3>("dependent v_add_f64", 1);
    run<4>("v_rcp_f64 + add (dependent)", 2);
    run<5>("2x permlane32_swap + add f64", 3);
    run<6>("2x dpp row_mirror mov + add f64", 3);
    run<7>("cvt f64->f32, fma f32, cvt back", 3);
    run<8>("sqrt(double) (dependent)", 0);
    run<9>("1.0/x + c IEEE (dependent)", 0);
    return 0;
}
