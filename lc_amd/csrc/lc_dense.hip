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

// T: element type of the network's maps (lc_map.h), TX: of the xyz map (T, or float next to 16-bit logits); the (B,N,.) rows are fp32
template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void lc_dense_frontend_fwd_kernel(const DenseParams p) {
    __shared__ float red[kWaves][2];
    const int b = blockIdx.x % p.B, slice = blockIdx.x / p.B, S = gridDim.x / p.B, tid = threadIdx.x;
    const int HW = p.H * p.W, n2 = 2 * HW;
    const T* __restrict__ lg = static_cast<const T*>(p.wlogits) + (size_t)b * p.wl_bs;
    const float lse = block_lse<kThreads>(lg, n2, red);  // joint softmax statistics over both channels (losses.py:355)
    if (slice == 0 && tid == 0) p.lse[b] = lse;
    const float scale = map_scalar_at(p.wscale, p.wscale_dtype, b);
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    const int chunk = (p.N + S - 1) / S, n0 = slice * chunk, n1 = min(p.N, n0 + chunk);
    for (int n = n0 + tid; n < n1; n += kThreads) {
        const int r = n / Wn, y = p.top + r * p.sample, x = p.left + (n - r * Wn) * p.sample, px = y * p.W + x;
        *reinterpret_cast<float2*>(p.pts2d + (ob + n) * 2) = make_float2((float)x, (float)y);
        *reinterpret_cast<float2*>(p.inv_std + (ob + n) * 2) =
            make_float2(__expf((float)lg[px] - lse) * scale, __expf((float)lg[HW + px] - lse) * scale);
        // test time: the segmentation mask of the sampled pixel, torch.sigmoid's own formula (full-precision exp, IEEE divide)
        if (p.vis_mask) p.vis_mask[ob + n] = (1.f / (1.f + expf(-(float)static_cast<const T*>(p.vis_logits)[(size_t)b * p.vis_bs + px]))) > p.vis_thresh ? 1 : 0;
    }
    if (p.xyz) {  // binary-code heads decode their 3D points with lc_bits_decode_gt_* instead
        const TX* __restrict__ xyz = static_cast<const TX*>(p.xyz) + (size_t)b * p.xyz_bs;
        const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                    ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
        float* __restrict__ o = p.pts3d + ob * 3;
        for (int e = 3 * n0 + tid; e < 3 * n1; e += kThreads) {  // one float per thread: the (N,3) rows are written fully coalesced
            const int n = e / 3, d = e - 3 * n, r = n / Wn;
            const int px = (p.top + r * p.sample) * p.W + p.left + (n - r * Wn) * p.sample;
            o[e] = (float)xyz[d * HW + px] * (d == 0 ? ns0 : (d == 1 ? ns1 : ns2));
        }
    }
}

// T: element type of the maps: the logits are read and BOTH gradients are written in it (lc_map.h)
template <bool VEC, typename T>
__global__ __launch_bounds__(kThreads) void lc_dense_frontend_bwd_kernel(const DenseBwdParams p) {
    __shared__ float red[kWaves];
    const int b = blockIdx.x % p.B, slice = blockIdx.x / p.B, S = gridDim.x / p.B, tid = threadIdx.x;
    const int HW = p.H * p.W;
    const T* __restrict__ lg = static_cast<const T*>(p.wlogits) + (size_t)b * p.wl_bs;
    const float lse = p.lse[b], scale = map_scalar_at(p.wscale, p.wscale_dtype, b);
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    // d_scale = sum_{n,c} p_nc * g_nc   (inv_std = p * scale); every slice forms the same sum in the same order
    float dot = 0.f;
    if (p.g_inv_std) {
        for (int n = tid; n < p.N; n += kThreads) {
            const int r = n / Wn, px = (p.top + r * p.sample) * p.W + p.left + (n - r * Wn) * p.sample;
            const float2 g = *reinterpret_cast<const float2*>(p.g_inv_std + (ob + n) * 2);
            dot += __expf((float)lg[px] - lse) * g.x + __expf((float)lg[HW + px] - lse) * g.y;
        }
    }
    dot = block_sum(dot, red);
    if (slice == 0 && tid == 0 && p.d_wscale) map_scalar_put(p.d_wscale, p.wscale_dtype, b, dot);
    const float sdot = scale * dot;
    const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
    constexpr int V = VEC ? 4 : 1;  // pixels per thread and iteration (VEC: W % 4 == 0, so the four share a row)
    const int units = HW / V, chunk = (units + S - 1) / S, u0 = slice * chunk, u1 = min(units, u0 + chunk);
    T* __restrict__ owl = p.d_wlogits ? static_cast<T*>(p.d_wlogits) + (size_t)b * 2 * HW : nullptr;
    T* __restrict__ oxyz = p.d_xyz ? static_cast<T*>(p.d_xyz) + (size_t)b * 3 * HW : nullptr;
    for (int u = u0 + tid; u < u1; u += kThreads) {
        const int px0 = u * V, y = px0 / p.W, x0 = px0 - y * p.W, dy = y - p.top;
        const bool row_hit = dy >= 0 && (dy % p.sample) == 0;
        const int nrow = row_hit ? (dy / p.sample) * Wn : 0;
        float l0[V], l1[V], o0[V], o1[V], gx[V], gy[V], gz[V];
        if (owl) {
            map_load<V>(lg + px0, l0);
            map_load<V>(lg + HW + px0, l1);
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
        constexpr bool kStream = VEC && LC_NT_GRAD_STORES;  // written once, read by the next kernel of the backward pass
        if (owl) {
            map_store<V>(owl + px0, o0, kStream);
            map_store<V>(owl + HW + px0, o1, kStream);
        }
        if (oxyz) {
            map_store<V>(oxyz + px0, gx, kStream);
            map_store<V>(oxyz + HW + px0, gy, kStream);
            map_store<V>(oxyz + 2 * HW + px0, gz, kStream);
        }
    }
}

inline int host_slices(int B) {
    const int s = (512 + B - 1) / B;
    return s < 1 ? 1 : (s > kMaxSlices ? kMaxSlices : s);
}

}  // namespace

int launch_dense_fwd(const DenseParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    const DenseParams p = with_dense_strides(p_in);
    if (p.xyz_dtype != p.map_dtype && p.xyz_dtype != kMapF32) return 2;
    LC_MAP_DISPATCH(p.map_dtype,
                    if (p.xyz_dtype == p.map_dtype) hipLaunchKernelGGL((lc_dense_frontend_fwd_kernel<T, T>), dim3(p.B * host_slices(p.B)), dim3(kThreads), 0, stream, p);
                    else hipLaunchKernelGGL((lc_dense_frontend_fwd_kernel<T, float>), dim3(p.B * host_slices(p.B)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_bwd(const DenseBwdParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    DenseBwdParams p = p_in;
    if (!p.wl_bs) p.wl_bs = 2ll * p.H * p.W;  // a dense batch
    const auto al = [&](const void* q) { return map_aligned4(q, p.map_dtype); };
    const bool vec = (p.W % 4) == 0 && (p.wl_bs % 4) == 0 && al(p.wlogits) && al(p.d_wlogits) && al(p.d_xyz);
    const dim3 grid(p.B * host_slices(p.B));
    LC_MAP_DISPATCH(p.map_dtype,
                    if (vec) hipLaunchKernelGGL((lc_dense_frontend_bwd_kernel<true, T>), grid, dim3(kThreads), 0, stream, p);
                    else hipLaunchKernelGGL((lc_dense_frontend_bwd_kernel<false, T>), grid, dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
