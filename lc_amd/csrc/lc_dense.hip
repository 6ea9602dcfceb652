// Dense-correspondence front end of the LC loss, forward and backward.
//
// Replaces the torch glue the reference runs between the network output and Loss_cov_mixed in the dense configs
// (losses.py:355-356 joint softmax over all 2*H*W weight logits x per-sample scale; losses.py:142-161 strided
// sub-sampling with phase (top,left), noc_scale multiply, (N,C) transposes) -- ~10 launches and their autograd twins --
// by one launch each way.  A sample is 32..128 KB of logits, a training batch 32..64 samples: one workgroup per sample
// would leave 3/4 of the chip idle, so S = ceil(512/B) (<= 8) workgroups share a sample.  Each of them recomputes the
// sample-wide reduction itself (same thread count and order -> bit-identical in all S; the re-reads hit L2 because
// blockIdx = slice*B + b puts the slices of a sample on the same XCD whenever B % 8 == 0) and then owns one slice of the
// outputs.  No workspace, no atomics, one launch.
//   fwd: lse over the 2HW logits (float4 stream, one online max/sum pass), then gather the slice's share of the
//        N = ceil((H-top)/s)*ceil((W-left)/s) sampled pixels:  pts2d[n] = (x,y);  inv_std[n,c] = exp(logit[c,y,x]-lse)*scale;
//        pts3d[n,d] = xyz[d,y,x]*noc_scale[d]
//   bwd: d_scale = sum p g;  d_logit = p*(scale*g[sampled] - scale*d_scale)  over ALL pixels;  d_xyz = d_pts3d*noc_scale scattered.
#include <cfloat>
#include <cstdint>

#include "lc_common.h"
#include "lc_dense_lse.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = kDenseLseThreads;
constexpr int kWaves = kThreads / kWave;
constexpr int kMaxSlices = 8;

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, kWave);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(kThreads) void lc_dense_frontend_fwd_kernel(const DenseParams p) {
    __shared__ float red[kWaves][2];
    const int b = blockIdx.x % p.B, slice = blockIdx.x / p.B, S = gridDim.x / p.B, tid = threadIdx.x;
    const int HW = p.H * p.W, n2 = 2 * HW;
    const float* __restrict__ lg = p.wlogits + (size_t)b * n2;
    const float lse = block_lse<kThreads>(lg, n2, red);  // joint softmax statistics over both channels (losses.py:355)
    if (slice == 0 && tid == 0) p.lse[b] = lse;
    const float scale = p.wscale[b];
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    const int chunk = (p.N + S - 1) / S, n0 = slice * chunk, n1 = min(p.N, n0 + chunk);
    for (int n = n0 + tid; n < n1; n += kThreads) {
        const int r = n / Wn, y = p.top + r * p.sample, x = p.left + (n - r * Wn) * p.sample, px = y * p.W + x;
        *reinterpret_cast<float2*>(p.pts2d + (ob + n) * 2) = make_float2((float)x, (float)y);
        *reinterpret_cast<float2*>(p.inv_std + (ob + n) * 2) =
            make_float2(__expf(lg[px] - lse) * scale, __expf(lg[HW + px] - lse) * scale);
        // test time: the segmentation mask of the sampled pixel, torch.sigmoid's own formula (full-precision exp, IEEE divide)
        if (p.vis_mask) p.vis_mask[ob + n] = (1.f / (1.f + expf(-p.vis_logits[(size_t)b * HW + px]))) > p.vis_thresh ? 1 : 0;
    }
    if (p.xyz) {  // binary-code heads decode their 3D points with lc_bits_decode_gt_* instead
        const float* __restrict__ xyz = p.xyz + (size_t)b * 3 * HW;
        const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                    ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
        float* __restrict__ o = p.pts3d + ob * 3;
        for (int e = 3 * n0 + tid; e < 3 * n1; e += kThreads) {  // one float per thread: the (N,3) rows are written fully coalesced
            const int n = e / 3, d = e - 3 * n, r = n / Wn;
            const int px = (p.top + r * p.sample) * p.W + p.left + (n - r * Wn) * p.sample;
            o[e] = xyz[d * HW + px] * (d == 0 ? ns0 : (d == 1 ? ns1 : ns2));
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void lc_dense_frontend_bwd_kernel(const DenseBwdParams p) {
    __shared__ float red[kWaves];
    const int b = blockIdx.x % p.B, slice = blockIdx.x / p.B, S = gridDim.x / p.B, tid = threadIdx.x;
    const int HW = p.H * p.W;
    const float* __restrict__ lg = p.wlogits + (size_t)b * 2 * HW;
    const float lse = p.lse[b], scale = p.wscale[b];
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    // d_scale = sum_{n,c} p_nc * g_nc   (inv_std = p * scale); every slice forms the same sum in the same order
    float dot = 0.f;
    if (p.g_inv_std) {
        for (int n = tid; n < p.N; n += kThreads) {
            const int r = n / Wn, px = (p.top + r * p.sample) * p.W + p.left + (n - r * Wn) * p.sample;
            const float2 g = *reinterpret_cast<const float2*>(p.g_inv_std + (ob + n) * 2);
            dot += __expf(lg[px] - lse) * g.x + __expf(lg[HW + px] - lse) * g.y;
        }
    }
    dot = block_sum(dot, red);
    if (slice == 0 && tid == 0 && p.d_wscale) p.d_wscale[b] = dot;
    const float sdot = scale * dot;
    const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
    constexpr int V = VEC ? 4 : 1;  // pixels per thread and iteration (VEC: W % 4 == 0, so the four share a row)
    const int units = HW / V, chunk = (units + S - 1) / S, u0 = slice * chunk, u1 = min(units, u0 + chunk);
    float* __restrict__ owl = p.d_wlogits ? p.d_wlogits + (size_t)b * 2 * HW : nullptr;
    float* __restrict__ oxyz = p.d_xyz ? p.d_xyz + (size_t)b * 3 * HW : nullptr;
    for (int u = u0 + tid; u < u1; u += kThreads) {
        const int px0 = u * V, y = px0 / p.W, x0 = px0 - y * p.W, dy = y - p.top;
        const bool row_hit = dy >= 0 && (dy % p.sample) == 0;
        const int nrow = row_hit ? (dy / p.sample) * Wn : 0;
        float l0[V], l1[V], o0[V], o1[V], gx[V], gy[V], gz[V];
        if (owl) {
            if constexpr (VEC) {
                const float4 a = *reinterpret_cast<const float4*>(lg + px0), c = *reinterpret_cast<const float4*>(lg + HW + px0);
                l0[0] = a.x; l0[1] = a.y; l0[2] = a.z; l0[3] = a.w;
                l1[0] = c.x; l1[1] = c.y; l1[2] = c.z; l1[3] = c.w;
            } else {
                l0[0] = lg[px0]; l1[0] = lg[HW + px0];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int dx = x0 + k - p.left;
            const bool hit = row_hit && dx >= 0 && (dx % p.sample) == 0;
            const int n = hit ? nrow + dx / p.sample : 0;
            if (owl) {
                float2 g = make_float2(0.f, 0.f);
                if (hit && p.g_inv_std) g = *reinterpret_cast<const float2*>(p.g_inv_std + (ob + n) * 2);
                o0[k] = __expf(l0[k] - lse) * (scale * g.x - sdot);  // softmax backward over the joint 2HW axis
                o1[k] = __expf(l1[k] - lse) * (scale * g.y - sdot);
            }
            gx[k] = gy[k] = gz[k] = 0.f;
            if (oxyz && hit && p.g_pts3d) {
                const float* g3 = p.g_pts3d + (ob + n) * 3;
                gx[k] = g3[0] * ns0; gy[k] = g3[1] * ns1; gz[k] = g3[2] * ns2;
            }
        }
        if constexpr (VEC) {
            if (owl) {
                store_stream4(owl + px0, o0[0], o0[1], o0[2], o0[3]);
                store_stream4(owl + HW + px0, o1[0], o1[1], o1[2], o1[3]);
            }
            if (oxyz) {
                store_stream4(oxyz + px0, gx[0], gx[1], gx[2], gx[3]);
                store_stream4(oxyz + HW + px0, gy[0], gy[1], gy[2], gy[3]);
                store_stream4(oxyz + 2 * HW + px0, gz[0], gz[1], gz[2], gz[3]);
            }
        } else {
            if (owl) { owl[px0] = o0[0]; owl[HW + px0] = o1[0]; }
            if (oxyz) { oxyz[px0] = gx[0]; oxyz[HW + px0] = gy[0]; oxyz[2 * HW + px0] = gz[0]; }
        }
    }
}

inline int host_slices(int B) {
    const int s = (512 + B - 1) / B;
    return s < 1 ? 1 : (s > kMaxSlices ? kMaxSlices : s);
}

}  // namespace

int launch_dense_fwd(const DenseParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_dense_frontend_fwd_kernel, dim3(p.B * host_slices(p.B)), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_bwd(const DenseBwdParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = (p.W % 4) == 0 && al(p.wlogits) && al(p.d_wlogits) && al(p.d_xyz);
    const dim3 grid(p.B * host_slices(p.B));
    if (vec) hipLaunchKernelGGL(lc_dense_frontend_bwd_kernel<true>, grid, dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL(lc_dense_frontend_bwd_kernel<false>, grid, dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
