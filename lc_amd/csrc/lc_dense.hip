// Dense-correspondence front end of the LC loss, forward and backward.
//
// Replaces the torch glue the reference runs between the network output and Loss_cov_mixed in the dense configs
// (losses.py:355-356 joint softmax over all 2*H*W weight logits x per-sample scale; losses.py:142-161 strided
// sub-sampling with phase (top,left), noc_scale multiply, (N,C) transposes) -- ~10 launches and their autograd twins --
// by one launch each way.  One workgroup per sample:
//   fwd: lse over the 2HW logits (float4 stream, online max/sum), then gather the N = ceil((H-top)/s)*ceil((W-left)/s)
//        sampled pixels:  pts2d[n] = (x,y);  inv_std[n,c] = exp(logit[c,y,x]-lse)*scale;  pts3d[n,d] = xyz[d,y,x]*noc_scale[d]
//   bwd: d_scale = sum p g;  d_logit = p*(scale*g[sampled] - scale*d_scale)  over ALL pixels;  d_xyz = d_pts3d*noc_scale scattered.
#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o = __shfl_xor(v, m, kWave);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();  // red may still be read from a previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kThreads) void lc_dense_frontend_fwd_kernel(const DenseParams p) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = p.H * p.W, n2 = 2 * HW;
    const float* lg = p.wlogits + (size_t)b * n2;
    // joint softmax statistics over both channels (losses.py:355)
    float mx = -INFINITY;
    for (int i = tid; i < n2; i += kThreads) mx = fmaxf(mx, lg[i]);
    mx = block_reduce(mx, red, true);
    float sm = 0.f;
    for (int i = tid; i < n2; i += kThreads) sm += __expf(lg[i] - mx);
    sm = block_reduce(sm, red, false);
    const float lse = mx + __logf(sm);
    if (tid == 0) p.lse[b] = lse;
    const float scale = p.wscale[b];
    const float* xyz = p.xyz + (size_t)b * 3 * HW;
    const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    for (int n = tid; n < p.N; n += kThreads) {
        const int y = p.top + (n / Wn) * p.sample, x = p.left + (n % Wn) * p.sample, px = y * p.W + x;
        *reinterpret_cast<float2*>(p.pts2d + (ob + n) * 2) = make_float2((float)x, (float)y);
        *reinterpret_cast<float2*>(p.inv_std + (ob + n) * 2) =
            make_float2(__expf(lg[px] - lse) * scale, __expf(lg[HW + px] - lse) * scale);
        if (p.xyz) {  // binary-code heads decode their 3D points with lc_bits_decode_gt_* instead
            float* o = p.pts3d + (ob + n) * 3;
            o[0] = xyz[px] * ns0; o[1] = xyz[HW + px] * ns1; o[2] = xyz[2 * HW + px] * ns2;
        }
    }
}

__global__ __launch_bounds__(kThreads) void lc_dense_frontend_bwd_kernel(const DenseBwdParams p) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = p.H * p.W;
    const float* lg = p.wlogits + (size_t)b * 2 * HW;
    const float lse = p.lse[b], scale = p.wscale[b];
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t ob = (size_t)b * p.N;
    // d_scale = sum_{n,c} p_nc * g_nc   (inv_std = p * scale)
    float dot = 0.f;
    if (p.g_inv_std) {
        for (int n = tid; n < p.N; n += kThreads) {
            const int y = p.top + (n / Wn) * p.sample, x = p.left + (n % Wn) * p.sample, px = y * p.W + x;
            const float2 g = *reinterpret_cast<const float2*>(p.g_inv_std + (ob + n) * 2);
            dot += __expf(lg[px] - lse) * g.x + __expf(lg[HW + px] - lse) * g.y;
        }
    }
    dot = block_reduce(dot, red, false);
    if (tid == 0 && p.d_wscale) p.d_wscale[b] = dot;
    const float sdot = scale * dot;
    const float ns0 = p.noc_scale ? p.noc_scale[3 * b] : 1.f, ns1 = p.noc_scale ? p.noc_scale[3 * b + 1] : 1.f,
                ns2 = p.noc_scale ? p.noc_scale[3 * b + 2] : 1.f;
    for (int px = tid; px < HW; px += kThreads) {
        const int y = px / p.W, x = px - y * p.W;
        const int dy = y - p.top, dx = x - p.left;
        const bool hit = dy >= 0 && dx >= 0 && (dy % p.sample) == 0 && (dx % p.sample) == 0;
        const int n = hit ? (dy / p.sample) * Wn + dx / p.sample : 0;
        if (p.d_wlogits) {
            float2 g = make_float2(0.f, 0.f);
            if (hit && p.g_inv_std) g = *reinterpret_cast<const float2*>(p.g_inv_std + (ob + n) * 2);
            float* o = p.d_wlogits + (size_t)b * 2 * HW;
            o[px] = __expf(lg[px] - lse) * (scale * g.x - sdot);            // softmax backward over the joint 2HW axis
            o[HW + px] = __expf(lg[HW + px] - lse) * (scale * g.y - sdot);
        }
        if (p.d_xyz) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (hit && p.g_pts3d) {
                const float* g3 = p.g_pts3d + (ob + n) * 3;
                gx = g3[0] * ns0; gy = g3[1] * ns1; gz = g3[2] * ns2;
            }
            float* o = p.d_xyz + (size_t)b * 3 * HW;
            o[px] = gx; o[HW + px] = gy; o[2 * HW + px] = gz;
        }
    }
}

}  // namespace

int launch_dense_fwd(const DenseParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_dense_frontend_fwd_kernel, dim3(p.B), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_bwd(const DenseBwdParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_dense_frontend_bwd_kernel, dim3(p.B), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
