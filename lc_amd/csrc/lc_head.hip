// Sparse keypoint head: spatial softmax + soft-argmax mean/std in ONE pass over the logits (HBM-bound).
//
// Replaces ptnet.py:59-66 (flatten -> softmax -> reshape) + ptnet.py:85-115 (softargmax_1d_cov x2, softargmax_2d_std):
// the reference materialises the (B,S,H,W) probability tensor and runs ~12 torch ops over it; here one workgroup owns
// one (H,W) map, keeps it in registers, stages exp(x-max) once through LDS for the row/column marginals, and writes
// 4 floats (+4 saved statistics).  Backward is a pure streaming kernel: with lse, mean and variance saved,
//   d/dlogit[h,w] = p[h,w] * (qx[w] + qy[h] - <p,q>),  qx[w] = g_mx w + g_vx (w-mx)^2,  <p,q> = g_mx mx + g_vx vx + (y terms)
// so the map is read once and the gradient written once: 3 x H x W x 4 bytes per map fwd+bwd in total.
#include <cstdint>

#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kHeadThreads = 256;

// Element types of the maps: fp32 (the reference's precision) and the 16-bit types a mixed-precision backbone emits
// (BASELINE.json configs 3 and 5).  Arithmetic is fp32 for all of them; 16-bit maps halve the HBM bytes of this
// bandwidth-bound kernel pair and the gradient is written back in the map's own type (round to nearest even).
template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<_Float16> { using type = uint2; };
template <> struct Vec4<__bf16> { using type = uint2; };

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// streaming (non-temporal) forms: a map that is read once / a gradient that is written once and not re-read by this kernel
template <typename T>
__device__ __forceinline__ float4 load4_nt(const T* q) {
    if constexpr (sizeof(T) == 4) {
        const v4f_t r = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(q));
        return make_float4(r.x, r.y, r.z, r.w);
    } else {
        const v2u_t r = __builtin_nontemporal_load(reinterpret_cast<const v2u_t*>(q));
        T h[4];
        __builtin_memcpy(h, &r, 8);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
}
template <typename T>
__device__ __forceinline__ void store4_nt(T* q, float4 v) {
    if constexpr (sizeof(T) == 4) {
        v4f_t r = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(r, reinterpret_cast<v4f_t*>(q));
    } else {
        const T h[4] = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        v2u_t r;
        __builtin_memcpy(&r, h, 8);
        __builtin_nontemporal_store(r, reinterpret_cast<v2u_t*>(q));
    }
}

// Memory policy of the fwd;bwd pair (HeadParams::variant bits), fixed from a sweep of all 16 combinations inside the alternating
// step on (256,64,64,64) maps (scripts/ubench/head_policy.py, profiles/r02/head_policy.txt):
//   bit0 non-temporal stores of the gradient: the 268 MB write-once stream no longer evicts the logits from the Infinity Cache /
//        leaves them dirty in L2 -- the forward that FOLLOWS a backward went from 73 us to 44 us (fp32), the step 163 -> 129 us;
//   bit2 backward walks the maps last-to-first: it starts on the lines the forward touched last (helps when the maps fit the
//        256 MiB Infinity Cache: bf16/fp16 86 -> 77 us per step; neutral for fp32);
//   bit1 / bit3 non-temporal LOADS (backward / forward) lose in every combination (the other kernel of the pair re-reads the map).
constexpr int kHeadPolicyF32 = 1, kHeadPolicy16 = 5;

template <typename T>
__device__ __forceinline__ float4 load4(const T* q) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const float4*>(q);
    } else {
        const uint2 r = *reinterpret_cast<const uint2*>(q);
        T h[4];
        __builtin_memcpy(h, &r, 8);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
}
template <typename T>
__device__ __forceinline__ void store4(T* q, float4 v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(q) = v;
    } else {
        const T h[4] = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        uint2 r;
        __builtin_memcpy(&r, h, 8);
        *reinterpret_cast<uint2*>(q) = r;
    }
}

// Eight consecutive elements per access: two 16-byte accesses for fp32 maps, ONE 16-byte access for the 16-bit types (an 8-byte
// access per lane moved 4.6-5.6 TB/s on bf16 maps where the 16-byte fp32 form moves 6.7 TB/s).
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void load8(const T* q, float (&x)[8], bool nt) {
    if constexpr (sizeof(T) == 4) {
        const float4 a = nt ? load4_nt<T>(q) : load4<T>(q), b = nt ? load4_nt<T>(q + 4) : load4<T>(q + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
        const v4u_t r = nt ? __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(q)) : *reinterpret_cast<const v4u_t*>(q);
        T h[8];
        __builtin_memcpy(h, &r, 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (float)h[j];
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* q, const float (&g)[8], bool nt) {
    if constexpr (sizeof(T) == 4) {
        if (nt) { store4_nt<T>(q, make_float4(g[0], g[1], g[2], g[3])); store4_nt<T>(q + 4, make_float4(g[4], g[5], g[6], g[7])); }
        else { store4<T>(q, make_float4(g[0], g[1], g[2], g[3])); store4<T>(q + 4, make_float4(g[4], g[5], g[6], g[7])); }
    } else {
        T h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (T)g[j];
        v4u_t r;
        __builtin_memcpy(&r, h, 16);
        if (nt) __builtin_nontemporal_store(r, reinterpret_cast<v4u_t*>(q));
        else *reinterpret_cast<v4u_t*>(q) = r;
    }
}

constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (1 ulp; no denormal range fix-up needed: x <= 0 here or the product with it underflows to 0 anyway)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// NV: vectors per thread, VEC: 4 (float4, needs W % 4 == 0) or 1
template <typename T, int NV, int VEC>
__global__ __launch_bounds__(kHeadThreads) void lc_head_fwd_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, W = p.W, HW = H * W, ld = W + 1;
    float* prob = smem;                   // [H][W+1] unnormalised exp
    float* px = smem + H * ld;            // [W]
    float* py = px + W;                   // [H]
    float* redf = py + H;                 // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t m = blockIdx.x;
    const T* in = static_cast<const T*>(p.in) + m * HW;

    float x[NV * VEC];
    float lmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int e = (tid + k * kHeadThreads) * VEC;
        if (e < HW) {
            if constexpr (VEC == 4) {
                const float4 v = load4<T>(in + e);
                x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
            } else {
                x[k] = (float)in[e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) x[VEC * k + j] = -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) lmax = fmaxf(lmax, x[VEC * k + j]);
    }
    float bmax;
    if (p.is_prob) {
        bmax = 0.f;
    } else {
        lmax = wave_max(lmax);
        if (lane == 0) redf[wave] = lmax;
        __syncthreads();
        bmax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    }
    float lsum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int e = (tid + k * kHeadThreads) * VEC;
        if (e < HW) {
            const int h = e / W, w = e - h * W;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float ex = p.is_prob ? x[VEC * k + j] : __expf(x[VEC * k + j] - bmax);
                prob[h * ld + w + j] = ex;
                lsum += ex;
            }
        }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) redf[4 + wave] = lsum;
    __syncthreads();
    const float bsum = (redf[4] + redf[5]) + (redf[6] + redf[7]);
    const float inv = p.is_prob ? 1.f : 1.f / bsum;
    // marginals (prob2d.sum(-2), prob2d.sum(-1), ptnet.py:107-108)
    for (int j = tid; j < W + H; j += kHeadThreads) {
        float s = 0.f;
        if (j < W) {
            for (int h = 0; h < H; ++h) s += prob[h * ld + j];
            px[j] = s * inv;
        } else {
            const int h = j - W;
            for (int w = 0; w < W; ++w) s += prob[h * ld + w];
            py[h] = s * inv;
        }
    }
    __syncthreads();
    // softargmax_1d_cov (ptnet.py:85-97): wave 0 -> x, wave 1 -> y
    if (wave < 2) {
        const float* pr = wave == 0 ? px : py;
        const int n = wave == 0 ? W : H;
        float mu = 0.f;
        for (int i = lane; i < n; i += kWave) mu += (float)i * pr[i];
        mu = wave_sum(mu);
        float var = 0.f;
        for (int i = lane; i < n; i += kWave) {
            const float d = (float)i - mu;
            var += d * d * pr[i];
        }
        var = wave_sum(var);
        if (lane == 0) {
            p.mean[m * 2 + wave] = mu;
            p.std[m * 2 + wave] = sqrtf(var + 1e-6f);
            p.stats[m * 4 + 1 + wave] = var;
            if (wave == 0) {
                p.stats[m * 4] = p.is_prob ? bsum : bmax + __logf(bsum);
                p.stats[m * 4 + 3] = 0.f;
            }
        }
    }
}


// ---- fast path: W = 4*LPR (64 or 128), H*W a multiple of 1024: marginals straight from registers -------------------
// Thread t owns float4 #(t + 256 k): row = (t + 256k) / LPR, columns 4*(t % LPR)..+3 (the same four columns for every k).
// A row lives in LPR consecutive lanes -> row sums by DPP (and one permlane16 swap when LPR = 32); a column lives in the
// lanes with equal (lane % LPR) -> column sums by permlane swaps; only 4 x (W + rows) floats cross waves through LDS.
template <int CTRL>
__device__ __forceinline__ float dpp_sum_step(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return x + __int_as_float(y);
}
__device__ __forceinline__ float row16_sum(float x) {  // all 16 lanes of a DPP row get the row total
    x = dpp_sum_step<0xB1>(x);   // quad_perm [1,0,3,2]
    x = dpp_sum_step<0x4E>(x);   // quad_perm [2,3,0,1]
    x = dpp_sum_step<0x141>(x);  // row_half_mirror
    x = dpp_sum_step<0x140>(x);  // row_mirror
    return x;
}
__device__ __forceinline__ float swap16_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(x), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float swap32_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(x), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}

template <typename T, int LPR, int NV>
__global__ __launch_bounds__(kHeadThreads) void lc_head_fwd_rows_kernel(const HeadParams p) {
    constexpr int W = 4 * LPR, H = NV * kHeadThreads / LPR, HW = H * W;
    __shared__ float colp[4][W];               // per-wave column partials
    __shared__ float py[H];
    __shared__ float px[W];
    __shared__ float redf[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t m = blockIdx.x;
    const T* in = static_cast<const T*>(p.in) + m * HW;
    const bool is_prob = p.is_prob != 0;

    float4 x[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) x[k] = load4<T>(in + 4 * (tid + k * kHeadThreads));
    float bmax = 0.f;
    if (!is_prob) {
        float lmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < NV; ++k) lmax = fmaxf(fmaxf(lmax, fmaxf(x[k].x, x[k].y)), fmaxf(x[k].z, x[k].w));
        lmax = wave_max(lmax);
        if (lane == 0) redf[wave] = lmax;
        __syncthreads();
        bmax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    }
    float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
    float rowsum[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float4 e = x[k];
        if (!is_prob) {
            e.x = __expf(e.x - bmax); e.y = __expf(e.y - bmax); e.z = __expf(e.z - bmax); e.w = __expf(e.w - bmax);
        }
        col.x += e.x; col.y += e.y; col.z += e.z; col.w += e.w;
        float r = (e.x + e.y) + (e.z + e.w);
        r = row16_sum(r);
        if (LPR == 32) r = swap16_sum(r);
        rowsum[k] = r;
    }
    // rows: lane (lane % LPR == 0) of each row group writes the un-normalised row sum
    if ((lane % LPR) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) py[(tid + k * kHeadThreads) / LPR] = rowsum[k];
    }
    // columns: combine the lanes of this wave that hold the same four columns
    if (LPR == 16) {
        col.x = swap16_sum(col.x); col.y = swap16_sum(col.y); col.z = swap16_sum(col.z); col.w = swap16_sum(col.w);
    }
    col.x = swap32_sum(col.x); col.y = swap32_sum(col.y); col.z = swap32_sum(col.z); col.w = swap32_sum(col.w);
    if (lane < LPR) *reinterpret_cast<float4*>(&colp[wave][4 * lane]) = col;
    __syncthreads();
    // px, total mass
    float bsum = 0.f;
    if (tid < W) {
        const float c = (colp[0][tid] + colp[1][tid]) + (colp[2][tid] + colp[3][tid]);
        px[tid] = c;
        bsum = c;
    }
    if (wave < W / kWave) {  // the waves that hold px entries
        bsum = wave_sum(bsum);
        if (lane == 0) redf[4 + wave] = bsum;
    }
    __syncthreads();
    bsum = W / kWave == 1 ? redf[4] : redf[4] + redf[5];
    const float inv = is_prob ? 1.f : 1.f / bsum;
    // softargmax_1d_cov (ptnet.py:85-97): wave 0 -> x from px, wave 1 -> y from py
    if (wave < 2) {
        const float* pr = wave == 0 ? px : py;
        constexpr int nx = W, ny = H;
        const int n = wave == 0 ? nx : ny;
        float mu = 0.f;
        for (int i = lane; i < n; i += kWave) mu += (float)i * (pr[i] * inv);
        mu = wave_sum(mu);
        float var = 0.f;
        for (int i = lane; i < n; i += kWave) {
            const float d = (float)i - mu;
            var += d * d * (pr[i] * inv);
        }
        var = wave_sum(var);
        if (lane == 0) {
            p.mean[m * 2 + wave] = mu;
            p.std[m * 2 + wave] = sqrtf(var + 1e-6f);
            p.stats[m * 4 + 1 + wave] = var;
            if (wave == 0) {
                p.stats[m * 4] = is_prob ? bsum : bmax + __logf(bsum);
                p.stats[m * 4 + 3] = 0.f;
            }
        }
    }
}

// ---- 64x64 maps (the sparse heads' shape): ONE WAVEFRONT PER MAP, no LDS, no barrier ---------------------------------
// Lane l = (row group g = l/8, column group c = l%8).  For k = 0..7 the wave reads 512 contiguous elements: lane l takes the
// eight elements 512k + 8l .. +7 = row 8k+g, columns 8c..8c+7 (16 or 32 contiguous bytes per lane, all 8-16 loads of the lane in
// flight at once).  Row sums reduce over the 8 lanes of a row group (DPP), column sums over the 8 row groups (DPP row_ror 8,
// permlane16/32 swaps); means and centred variances come from those marginals exactly as ptnet.py:85-97 forms them.
template <int CTRL>
__device__ __forceinline__ float dpp_max_step(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return fmaxf(x, __int_as_float(y));
}
__device__ __forceinline__ float group8_sum(float x) {  // over lanes differing in bits 0..2
    x = dpp_sum_step<0xB1>(x);   // quad_perm [1,0,3,2]  (xor 1)
    x = dpp_sum_step<0x4E>(x);   // quad_perm [2,3,0,1]  (xor 2)
    x = dpp_sum_step<0x141>(x);  // row_half_mirror      (pairs the two quads of an 8-lane half row)
    return x;
}
__device__ __forceinline__ float across_groups_sum(float x) {  // over lanes differing in bits 3..5
    x = dpp_sum_step<0x128>(x);  // row_ror 8  (xor 8 within a 16-lane row)
    x = swap16_sum(x);
    x = swap32_sum(x);
    return x;
}

typedef float v2f_t __attribute__((ext_vector_type(2)));

// PROB (the probabilities-in form, ptnet.softargmax_2d_std on its own) is a template flag: as a run-time one it cost a select per
// element.  The element loop works on PAIRS (v_pk_fma_f32 / v_pk_add_f32: two fp32 lanes per issue slot): with 16-bit maps the
// forward is VALU-bound, not HBM-bound (27.5 us for 134 MB before this form), and an element now costs the conversion, half a
// max3, half an fma, the v_exp_f32 and two half adds.
template <typename T, bool PROB>
__global__ __launch_bounds__(kHeadThreads) void lc_head_fwd_wave64_kernel(const HeadParams p) {
    constexpr int S = 64, HW = S * S;
    const int lane = threadIdx.x & 63;
    const size_t m = (size_t)blockIdx.x * (kHeadThreads / kWave) + (threadIdx.x >> 6);
    if (m >= (size_t)p.M) return;
    const T* in = static_cast<const T*>(p.in) + m * HW + 8 * lane;
    constexpr bool is_prob = PROB;
    const int g = lane >> 3, c = lane & 7;
    const bool ntl = (p.variant & 8) != 0;

    float x[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) load8<T>(in + 512 * k, x[k], ntl);
    float bmax = 0.f;
    if constexpr (!PROB) {
        float lmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) lmax = fmaxf(lmax, x[k][j]);
        bmax = wave_max(lmax);
    }
    const float nbmax = -bmax * kLog2e;
    v2f_t col2[4];
    float row[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) col2[j] = v2f_t{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v2f_t r2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v2f_t e = {x[k][2 * j], x[k][2 * j + 1]};
            if constexpr (!PROB) {  // exp(x - max): one (packed) fma + v_exp_f32
                const v2f_t t = __builtin_elementwise_fma(e, v2f_t{kLog2e, kLog2e}, v2f_t{nbmax, nbmax});
                e = v2f_t{exp2_fast(t.x), exp2_fast(t.y)};
            }
            col2[j] += e;
            r2 += e;
        }
        row[k] = group8_sum(r2.x + r2.y);  // un-normalised mass of row 8k+g, in all 8 lanes of the group
    }
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col2[j] = v2f_t{across_groups_sum(col2[j].x), across_groups_sum(col2[j].y)};  // un-normalised mass of columns 8c+2j, +1, in every row group
        tot += col2[j].x + col2[j].y;
    }
    const float bsum = group8_sum(tot);
    const float inv = is_prob ? 1.f : 1.f / bsum;
    // softargmax_1d_cov (ptnet.py:85-97) on the two marginals, two coordinates per instruction
    const float cx = (float)(8 * c), gy = (float)g;
    v2f_t wx[4], wy[4], row2[4], mx2 = {0.f, 0.f}, my2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wx[j] = v2f_t{cx, cx} + v2f_t{(float)(2 * j), (float)(2 * j + 1)};
        wy[j] = v2f_t{gy, gy} + v2f_t{(float)(16 * j), (float)(16 * j + 8)};
        col2[j] *= v2f_t{inv, inv};
        row2[j] = v2f_t{row[2 * j], row[2 * j + 1]} * v2f_t{inv, inv};
        mx2 = __builtin_elementwise_fma(wx[j], col2[j], mx2);
        my2 = __builtin_elementwise_fma(wy[j], row2[j], my2);
    }
    const float mx = group8_sum(mx2.x + mx2.y);
    const float my = across_groups_sum(my2.x + my2.y);
    v2f_t vx2 = {0.f, 0.f}, vy2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f_t dx = wx[j] - v2f_t{mx, mx}, dy = wy[j] - v2f_t{my, my};
        vx2 = __builtin_elementwise_fma(dx * dx, col2[j], vx2);
        vy2 = __builtin_elementwise_fma(dy * dy, row2[j], vy2);
    }
    const float vx = group8_sum(vx2.x + vx2.y);
    const float vy = across_groups_sum(vy2.x + vy2.y);
    if (lane == 0) {
        *reinterpret_cast<float2*>(p.mean + m * 2) = make_float2(mx, my);
        *reinterpret_cast<float2*>(p.std + m * 2) = make_float2(sqrtf(vx + 1e-6f), sqrtf(vy + 1e-6f));
        *reinterpret_cast<float4*>(p.stats + m * 4) = make_float4(is_prob ? bsum : bmax + __logf(bsum), vx, vy, 0.f);
    }
}

template <typename T, int VEC, bool COLFIX = false>
__global__ __launch_bounds__(kHeadThreads) void lc_head_bwd_kernel(const HeadBwdParams p) {
    // every fused multiply-add of this kernel is written out and contraction is off: the VEC = 1 / 4 (fp32) and VEC = 8 (16-bit)
    // instantiations must produce the same bits -- the 16-bit gradient is DEFINED as the fp32 one rounded to nearest even
#pragma clang fp contract(off)
    const int H = p.H, W = p.W, HW = H * W;
    const size_t m = (p.variant & 4) ? (size_t)(p.M - 1) - blockIdx.x : (size_t)blockIdx.x;
    const bool nts = (p.variant & 1) != 0, ntl = (p.variant & 2) != 0;
    const T* in = static_cast<const T*>(p.in) + m * HW;
    T* out = static_cast<T*>(p.g_in) + m * HW;
    const float mx = p.mean[m * 2], my = p.mean[m * 2 + 1];
    const float sx = p.std[m * 2], sy = p.std[m * 2 + 1];
    const float s0 = p.stats[m * 4], vx = p.stats[m * 4 + 1], vy = p.stats[m * 4 + 2];
    const float gmx = p.g_mean[m * 2], gmy = p.g_mean[m * 2 + 1];
    const float gvx = p.g_std[m * 2] / (2.f * sx), gvy = p.g_std[m * 2 + 1] / (2.f * sy);  // d sqrt(var+1e-6)
    const bool is_prob = p.is_prob != 0;
    // logits: <p,q>;  prob input: -2 g_v m (1 - sum p) multiplies the coordinate (d var/d mean term, ptnet.py:94-96)
    const float cdot = gmx * mx + gmy * my + gvx * vx + gvy * vy;
    const float ex = is_prob ? -2.f * gvx * mx * (1.f - s0) : 0.f;
    const float ey = is_prob ? -2.f * gvy * my * (1.f - s0) : 0.f;
    [[maybe_unused]] const float cdot_l = is_prob ? 0.f : cdot;
    if constexpr (COLFIX) {
        // (kHeadThreads * VEC) % W == 0: a thread meets the same VEC columns in every iteration, so the column part of
        //   q[h,w] = gmx w + gvx (w - mx)^2 + ex w  +  qy[h]
        // is formed once per thread and an element costs: fma + v_exp (exp(x - lse)), one add, one mul (+ the conversion for 16-bit maps)
        // -- the generic loop below spends ~12 VALU instructions per element, which made the 16-bit backward VALU-bound (53 us for
        // 268 MB = 5.0 TB/s) instead of HBM-bound.
        const int e0 = threadIdx.x * VEC, w0 = e0 % W;
        float qx[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float wf = (float)(w0 + j), dx = wf - mx;
            qx[j] = __builtin_fmaf(gvx * dx, dx, __builtin_fmaf(gmx, wf, -cdot_l));
        }
        const float ns0 = -s0 * kLog2e;
        const int rows_per_iter = kHeadThreads * VEC / W;
        int h = e0 / W;
        for (int e = e0; e < HW; e += kHeadThreads * VEC, h += rows_per_iter) {
            const float dy = (float)h - my;
            const float qy = __builtin_fmaf(gvy * dy, dy, gmy * (float)h);
            float xin[VEC], g[VEC];
            if constexpr (VEC == 8) load8<T>(in + e, xin, ntl);
            else {
                const float4 v = ntl ? load4_nt<T>(in + e) : load4<T>(in + e);
                xin[0] = v.x; xin[1] = v.y; xin[2] = v.z; xin[3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] = exp2_fast(__builtin_fmaf(xin[j], kLog2e, ns0)) * (qx[j] + qy);  // (logits form: ex = ey = 0)
            if constexpr (VEC == 8) store8<T>(out + e, g, nts);
            else if (nts) store4_nt<T>(out + e, make_float4(g[0], g[1], g[2], g[3]));
            else store4<T>(out + e, make_float4(g[0], g[1], g[2], g[3]));
        }
        return;
    }
    // generic form (any shape, probability input): explicit fused multiply-adds like the fast path
    for (int e = threadIdx.x * VEC; e < HW; e += kHeadThreads * VEC) {
        const int h = e / W, w0 = e - h * W;
        const float dy = (float)h - my, hf = (float)h;
        const float qy = __builtin_fmaf(gvy * dy, dy, __builtin_fmaf(gmy, hf, ey * hf));
        float xin[VEC], g[VEC];
        if constexpr (VEC == 8) {
            load8<T>(in + e, xin, ntl);
        } else if constexpr (VEC == 4) {
            const float4 v = ntl ? load4_nt<T>(in + e) : load4<T>(in + e);
            xin[0] = v.x; xin[1] = v.y; xin[2] = v.z; xin[3] = v.w;
        } else {
            xin[0] = (float)in[e];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float wf = (float)(w0 + j);
            const float dx = wf - mx;
            const float q = __builtin_fmaf(gvx * dx, dx, __builtin_fmaf(gmx, wf, __builtin_fmaf(ex, wf, qy)));
            g[j] = is_prob ? q : __expf(xin[j] - s0) * (q - cdot);
        }
        if constexpr (VEC == 8) {
            store8<T>(out + e, g, nts);
        } else if constexpr (VEC == 4) {
            if (nts) store4_nt<T>(out + e, make_float4(g[0], g[1], g[2], g[3]));
            else store4<T>(out + e, make_float4(g[0], g[1], g[2], g[3]));
        } else {
            out[e] = (T)g[0];
        }
    }
}

template <typename T, int VEC>
int launch_fwd_nv(const HeadParams& p, hipStream_t stream, int nv, size_t smem) {
#define LC_HEAD_CASE(NVV)                                                                                       \
    case NVV:                                                                                                   \
        if (smem > 48 * 1024)                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lc_head_fwd_kernel<T, NVV, VEC>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                   \
        hipLaunchKernelGGL((lc_head_fwd_kernel<T, NVV, VEC>), dim3(p.M), dim3(kHeadThreads), smem, stream, p);   \
        break;
    switch (nv) {
        LC_HEAD_CASE(1) LC_HEAD_CASE(2) LC_HEAD_CASE(4) LC_HEAD_CASE(8) LC_HEAD_CASE(16) LC_HEAD_CASE(32)
        default: return 3;
    }
#undef LC_HEAD_CASE
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

template <typename T>
int launch_head_fwd_t(const HeadParams& p, hipStream_t stream) {
    const int HW = p.H * p.W;
    const uintptr_t vec_mask = 4 * sizeof(T) - 1;  // four elements per access: 16 bytes (fp32) or 8 bytes (16-bit maps)
    const bool vec4 = (p.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & vec_mask) == 0);
    const int vec = vec4 ? 4 : 1;
    int nv = (HW + kHeadThreads * vec - 1) / (kHeadThreads * vec);
    int nvp = 1;
    while (nvp < nv) nvp <<= 1;
    const bool out8 = ((reinterpret_cast<uintptr_t>(p.mean) | reinterpret_cast<uintptr_t>(p.std)) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.stats) & 15) == 0;
    if (vec4 && out8 && p.W == 64 && p.H == 64 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0) {  // 16-byte accesses (load8)
        const int waves = kHeadThreads / kWave;
        if (p.is_prob) hipLaunchKernelGGL((lc_head_fwd_wave64_kernel<T, true>), dim3((p.M + waves - 1) / waves), dim3(kHeadThreads), 0, stream, p);
        else hipLaunchKernelGGL((lc_head_fwd_wave64_kernel<T, false>), dim3((p.M + waves - 1) / waves), dim3(kHeadThreads), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (vec4 && p.W == 64 && p.H == 64) {
        hipLaunchKernelGGL((lc_head_fwd_rows_kernel<T, 16, 4>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (vec4 && p.W == 128 && p.H == 128) {
        hipLaunchKernelGGL((lc_head_fwd_rows_kernel<T, 32, 16>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    const size_t smem = sizeof(float) * ((size_t)p.H * (p.W + 1) + p.W + p.H + 8);
    if (smem > 160 * 1024 || nvp > 32) return 3;  // map too large for the single-pass design
    return vec4 ? launch_fwd_nv<T, 4>(p, stream, nvp, smem) : launch_fwd_nv<T, 1>(p, stream, nvp, smem);
}

template <typename T>
int launch_head_bwd_t(const HeadBwdParams& p, hipStream_t stream) {
    const uintptr_t vec_mask = 4 * sizeof(T) - 1;
    const bool vec4 = (p.W % 4 == 0) && (((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.g_in)) & vec_mask) == 0);
    const bool vec8 = sizeof(T) == 2 && (p.W % 8 == 0) && (((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.g_in)) & 15) == 0);
    const bool logits = p.is_prob == 0;  // the column-fixed fast path covers the logits form (every training call site)
    if (vec8 && logits && (kHeadThreads * 8) % p.W == 0)
        hipLaunchKernelGGL((lc_head_bwd_kernel<T, 8, true>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
    else if (vec8)  // 16-bit maps: one 16-byte access per eight elements
        hipLaunchKernelGGL((lc_head_bwd_kernel<T, 8>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
    else if (vec4 && logits && (kHeadThreads * 4) % p.W == 0)
        hipLaunchKernelGGL((lc_head_bwd_kernel<T, 4, true>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
    else if (vec4)
        hipLaunchKernelGGL((lc_head_bwd_kernel<T, 4>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
    else
        hipLaunchKernelGGL((lc_head_bwd_kernel<T, 1>), dim3(p.M), dim3(kHeadThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace

int launch_head_fwd(const HeadParams& p_in, hipStream_t stream) {
    if (p_in.M <= 0) return 0;
    HeadParams p = p_in;
    p.variant = p.dtype == kHeadF32 ? kHeadPolicyF32 : kHeadPolicy16;
    switch (p.dtype) {
        case kHeadF32: return launch_head_fwd_t<float>(p, stream);
        case kHeadF16: return launch_head_fwd_t<_Float16>(p, stream);
        case kHeadBF16: return launch_head_fwd_t<__bf16>(p, stream);
        default: return 4;
    }
}

int launch_head_bwd(const HeadBwdParams& p_in, hipStream_t stream) {
    if (p_in.M <= 0) return 0;
    HeadBwdParams p = p_in;
    p.variant = p.dtype == kHeadF32 ? kHeadPolicyF32 : kHeadPolicy16;
    switch (p.dtype) {
        case kHeadF32: return launch_head_bwd_t<float>(p, stream);
        case kHeadF16: return launch_head_bwd_t<_Float16>(p, stream);
        case kHeadBF16: return launch_head_bwd_t<__bf16>(p, stream);
        default: return 4;
    }
}

}  // namespace lc
