// Test helper (NOT part of the product library): a kernel that does nothing but hold compute units for a given wall-clock time, so that
// tests can take a known share of the chip away from the launches under test (tests/test_gpu_contention.py).
// occupy(blocks, threads, lds_bytes, ticks, stream): `blocks` workgroups of `threads` threads, each with `lds_bytes` of LDS, spin until
// `ticks` of the 100 MHz constant clock (s_memrealtime) have passed since their own start.  1024-thread workgroups are 16 of a compute
// unit's 32 wave slots: 2 x CUs of them fill the chip.
#include <hip/hip_runtime.h>

namespace {
__global__ __launch_bounds__(1024) void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
    if (ticks < 0) sink[threadIdx.x] = lds[threadIdx.x];  // never: keeps the LDS allocation alive
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int occupy(int blocks, int threads, int lds_bytes, long long ticks, void* stream) {
    if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0) return 1;
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return 2;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, static_cast<hipStream_t>(stream), ticks, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
