// Shared device helpers for the lc_amd HIP kernels (gfx950 / CDNA4, wave64).
//
// Cross-lane reductions here never touch LDS: the 32- and 16-lane exchanges use gfx950's v_permlane32_swap /
// v_permlane16_swap (a swap of register halves IS the reduce-scatter exchange, so no select is needed), the 8-, 4-, 2-
// and 1-lane exchanges use DPP row_mirror / row_half_mirror / quad_perm moves.
#pragma once
#include <hip/hip_runtime.h>

namespace lc {

constexpr int kWave = 64;

// DPP controls (cdna4 ISA 'DPP_CTRL')
constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppIdentity = 0xE4;      // quad_perm:[0,1,2,3]
constexpr int kDppRowMirror = 0x140;    // lane i <-> 15-i inside each row of 16
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7-i inside each half row of 8

template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// value of `v` in lane `src` (ds_bpermute: LDS crossbar, no LDS memory)
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const int lo = __shfl(__double2loint(v), src, kWave), hi = __shfl(__double2hiint(v), src, kWave);
    return __hiloint2double(hi, lo);
}

// a' (returned in a) and b' after swapping a's upper 32 lanes with b's lower 32 lanes; a'+b' then holds, in the lower
// half-wave, the two-half sum of a and, in the upper half-wave, the two-half sum of b.
__device__ __forceinline__ double swap32_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// same for rows of 16: even rows end with the (row, row+1) sum of a, odd rows with that of b
__device__ __forceinline__ double swap16_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// lanes with bit 3 set ("up", banks 2-3 of each row) keep hi, the others keep lo; the partner (row mirror) supplies
// the same quantity.  BANK_UP = bank mask of the up lanes: 0xC for the 8-exchange, 0xA for the 4-exchange.
template <int CTRL, int BANK_UP>
__device__ __forceinline__ double dpp_exchange_add(double lo, double hi) {
    constexpr int BANK_DOWN = 0xF & ~BANK_UP;
    const double keep = dpp_mov_f64<kDppIdentity, BANK_UP>(lo, hi);              // up lanes <- hi
    double recv = dpp_mov_f64<CTRL, BANK_UP>(lo, hi);                            // up lanes <- partner's hi
    recv = dpp_mov_f64<CTRL, BANK_DOWN>(recv, lo);                               // down lanes <- partner's lo
    return keep + recv;
}

// All-reduce of K doubles across the 64 lanes of a wave (every lane gets the sum), LDS-free.
template <int K>
__device__ __forceinline__ void wave_allreduce(double (&v)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double x = v[i];
        x += dpp_mov_f64<kDppQuadXor1>(x, x);
        x += dpp_mov_f64<kDppQuadXor2>(x, x);
        x += dpp_mov_f64<kDppHalfMirror>(x, x);
        x += dpp_mov_f64<kDppRowMirror>(x, x);
        x = swap16_add(x, x);
        x = swap32_add(x, x);
        v[i] = x;
    }
}

// Reduce-scatter of K = 16*R doubles across a wave: after the call lane l holds, in v[0..R-1], the full
// wave sums of entries base..base+R-1 with base = R*(8*b5 + 4*b4 + 2*b3 + b2) (b_i = bit i of l).
// K/2 + K/4 + K/8 + K/16 + 2R exchanges instead of 6K for a butterfly all-reduce, none of them through LDS.
template <int K>
__device__ __forceinline__ void wave_reduce_scatter16(double (&v)[K], int /*lane*/) {
    static_assert(K % 16 == 0, "K must be a multiple of 16");
    constexpr int R = K / 16;
#pragma unroll
    for (int i = 0; i < K / 2; ++i) v[i] = swap32_add(v[i], v[i + K / 2]);
#pragma unroll
    for (int i = 0; i < K / 4; ++i) v[i] = swap16_add(v[i], v[i + K / 4]);
#pragma unroll
    for (int i = 0; i < K / 8; ++i) v[i] = dpp_exchange_add<kDppRowMirror, 0xC>(v[i], v[i + K / 8]);
#pragma unroll
    for (int i = 0; i < K / 16; ++i) v[i] = dpp_exchange_add<kDppHalfMirror, 0xA>(v[i], v[i + K / 16]);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        v[i] += dpp_mov_f64<kDppQuadXor2>(v[i], v[i]);
        v[i] += dpp_mov_f64<kDppQuadXor1>(v[i], v[i]);
    }
}

__device__ __forceinline__ int scatter16_base(int lane, int R) {
    return R * (((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1));
}

// 1/x to full double precision without the IEEE division sequence (v_rcp_f64 + two Newton steps; x finite, non-zero,
// normal -- every use below is a pivot, a depth or a norm that is checked separately)
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// index of (i,j), i<=j, in a packed upper-triangular 6x6 (21 entries, row-major)
__host__ __device__ constexpr int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

}  // namespace lc
