// Shared device helpers for the lc_amd HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace lc {

constexpr int kWave = 64;

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    // 64-bit cross-lane exchange as two 32-bit ds_bpermute/DPP moves
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, kWave);
    hi = __shfl_xor(hi, mask, kWave);
    return __hiloint2double(hi, lo);
}

// All-reduce of K doubles across the 64 lanes of a wave (K small: butterfly, every lane gets the sum).
template <int K>
__device__ __forceinline__ void wave_allreduce(double (&v)[K]) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int i = 0; i < K; ++i) v[i] += shfl_xor_f64(v[i], m);
    }
}

// Reduce-scatter of K = 16*R doubles across a wave: after the call lane l holds, in v[0..R-1], the full
// wave sums of entries base..base+R-1 with base = R*(8*b5 + 4*b4 + 2*b3 + b2) (b_i = bit i of l).
// Cost: K/2 + K/4 + K/8 + K/16 + 2R exchanges instead of 6K for a butterfly all-reduce.
template <int K, int W, int M>
struct ReduceScatterStep {
    static __device__ __forceinline__ void run(double (&v)[K], int lane) {
        constexpr int half = W / 2;
        const bool up = (lane & M) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const double send = up ? v[i] : v[i + half];
            const double keep = up ? v[i + half] : v[i];
            v[i] = keep + shfl_xor_f64(send, M);
        }
        if constexpr (M > 4) ReduceScatterStep<K, half, M / 2>::run(v, lane);
    }
};

template <int K>
__device__ __forceinline__ void wave_reduce_scatter16(double (&v)[K], int lane) {
    static_assert(K % 16 == 0, "K must be a multiple of 16");
    constexpr int R = K / 16;
    ReduceScatterStep<K, K, 32>::run(v, lane);
#pragma unroll
    for (int m = 2; m >= 1; m >>= 1) {
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] += shfl_xor_f64(v[i], m);
    }
}

__device__ __forceinline__ int scatter16_base(int lane, int R) {
    return R * (((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1));
}

// index of (i,j), i<=j, in a packed upper-triangular 6x6 (21 entries, row-major)
__host__ __device__ constexpr int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

}  // namespace lc
