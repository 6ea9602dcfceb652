// The dense heads' auxiliary losses of Loss_fn.forward (losses.py:281-316), forward and backward, one launch each way:
//   loss_noc        = mean |xyz_noc * msk_noc - xyz_noc_tgt|                          (F.l1_loss, losses.py:293-295)
//   loss_seg        = mean seg(msk_vis_logits, msk_vis)                               (losses.py:296)
//   loss_weight_seg = mean seg(xyz_weight_logits, msk_vis broadcast over 2 channels)  (warm-up blend, losses.py:303-306)
// with seg = F.binary_cross_entropy_with_logits or Loss_seg_L1 (|sigmoid(x) - t|, losses.py:219-236).
// The reference spends ~25 element-wise / reduction launches and their autograd twins on these (mul, sub, abs, mean, sigmoid, sign,
// fills, scalar multiplies); the maps of a batch are a few MB, so every one of those launches is pure latency.  Here: a grid-stride
// pass over the pixels with the three sums in double precision, per-block partials in a fixed order, summed by the last block to
// arrive (deterministic, the pattern of lc_clip.hip); the backward pass is one element-wise launch that reads the three upstream
// cotangents from device scalars (no host synchronisation, hipGraph-replayable).
#include <algorithm>

#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }  // torch.sigmoid's formula
// F.binary_cross_entropy_with_logits (ATen Loss.cpp): (1 - t) x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float seg_loss(float x, float t, int type) {
    if (type == 0) return (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
    return fabsf(sigmoidf_(x) - t);
}
__device__ __forceinline__ float seg_grad(float x, float t, int type) {
    const float s = sigmoidf_(x);
    if (type == 0) return s - t;
    const float d = s - t;
    return (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (s * (1.f - s));  // sign(s - t) * sigmoid'(x)
}

__device__ __forceinline__ void block_sum3(double (&v)[3], double (*red)[3]) {
    wave_allreduce<3>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 3; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

__device__ __forceinline__ float mask_at(const DenseAuxParams& p, size_t i) {
    return p.msk_noc_u8 ? (p.msk_noc_u8[i] ? 1.f : 0.f) : p.msk_noc_f32[i];
}

__global__ __launch_bounds__(kThreads) void lc_dense_aux_fwd_kernel(const DenseAuxParams p) {
    __shared__ double red[4][3];
    __shared__ bool last;
    const long long n = (long long)p.B * p.HW;
    double acc[3] = {0, 0, 0};
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const long long b = i / p.HW, px = i - b * p.HW;
        const float t = p.msk_vis[i];
        if (p.xyz) {
            const float m = mask_at(p, (size_t)i);
            float s = 0.f;
            for (int c = 0; c < 3; ++c) {
                const size_t e = ((size_t)b * 3 + c) * p.HW + px;
                s += fabsf(p.xyz[e] * m - p.noc_tgt[e]);
            }
            acc[0] += (double)s;
        }
        acc[1] += (double)seg_loss(p.seg_logits[i], t, p.seg_type);
        if (p.wlogits) {
            const size_t e = (size_t)b * 2 * p.HW + px;
            acc[2] += (double)(seg_loss(p.wlogits[e], t, p.seg_type) + seg_loss(p.wlogits[e + p.HW], t, p.seg_type));
        }
    }
    block_sum3(acc, red);
    if (threadIdx.x == 0) {
        // the block's partials are written through the caches and acknowledged before the block counts itself in; the last block
        // reads all of them around the caches: no agent-scope fence (an L2 write-back per block, ~10 us for 512 blocks: lc_common.h)
        for (int k = 0; k < 3; ++k) xcd_store(p.partials + 3 * blockIdx.x + k, acc[k]);
        xcd_stores_done();
        last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    double s[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads)
        for (int k = 0; k < 3; ++k) s[k] += xcd_load(p.partials + 3 * i + k);
    block_sum3(s, red);
    if (threadIdx.x == 0) {
        p.losses[0] = p.xyz ? (float)(s[0] / (3.0 * (double)n)) : 0.f;
        p.losses[1] = (float)(s[1] / (double)n);
        p.losses[2] = p.wlogits ? (float)(s[2] / (2.0 * (double)n)) : 0.f;
        xcd_store(p.ticket, 0u);  // ready for the next call on this stream
    }
}

__global__ __launch_bounds__(kThreads) void lc_dense_aux_bwd_kernel(const DenseAuxParams p) {
    const long long n = (long long)p.B * p.HW;
    const float g0 = (p.g_noc && p.d_xyz) ? *p.g_noc / (3.f * (float)n) : 0.f;
    const float g1 = p.g_seg ? *p.g_seg / (float)n : 0.f;
    const float g2 = (p.g_wseg && p.d_wlogits) ? *p.g_wseg / (2.f * (float)n) : 0.f;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const long long b = i / p.HW, px = i - b * p.HW;
        const float t = p.msk_vis[i];
        if (p.d_xyz) {
            const float m = mask_at(p, (size_t)i);
            for (int c = 0; c < 3; ++c) {
                const size_t e = ((size_t)b * 3 + c) * p.HW + px;
                const float d = p.xyz[e] * m - p.noc_tgt[e];
                p.d_xyz[e] = (d > 0.f ? g0 : (d < 0.f ? -g0 : 0.f)) * m;  // torch.sign(0) = 0
            }
        }
        if (p.d_seg) p.d_seg[i] = g1 * seg_grad(p.seg_logits[i], t, p.seg_type);
        if (p.d_wlogits) {
            const size_t e = (size_t)b * 2 * p.HW + px;
            p.d_wlogits[e] = g2 * seg_grad(p.wlogits[e], t, p.seg_type);
            p.d_wlogits[e + p.HW] = g2 * seg_grad(p.wlogits[e + p.HW], t, p.seg_type);
        }
    }
}

int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    return (int)(g < 1 ? 1 : (g > kDenseAuxMaxBlocks ? kDenseAuxMaxBlocks : g));
}

}  // namespace

int launch_dense_aux_fwd(const DenseAuxParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    // four pixels per thread up to one block per compute unit: every block ends with one counted arrival
    const long long n = (long long)p.B * p.HW;
    const int grid = (int)std::min<long long>(256, std::max<long long>(1, (n + 4 * kThreads - 1) / (4 * kThreads)));
    hipLaunchKernelGGL(lc_dense_aux_fwd_kernel, dim3(grid), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_aux_bwd(const DenseAuxParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_dense_aux_bwd_kernel, dim3(grid_for((long long)p.B * p.HW)), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
