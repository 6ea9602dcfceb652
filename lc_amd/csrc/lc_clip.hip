// EMA-adaptive global-norm gradient clipping (lib/utils/grad.py:5-30 NormClipper + :33-83 clip_norm), the backward hook the
// dense heads hang on their weight logits / weight scale / 3D points (losses.py:245-247,343-352,378-381).
// The reference spends ~15 small torch launches per hook call (norm, add, div, clamp, mul, the EMA update); here:
//   lc_sqnorm_kernel      sum of squares of the gradient -> one device float (deterministic: per-block partials in a
//                         fixed order, summed by the last block to arrive), which also snapshots max_norm so that the
//                         second kernel can overwrite it IN PLACE (static addresses: the pair replays inside a hipGraph)
//   [ all-reduce of that float over the process group when the batch is sharded: SURVEY.md 8e ]
//   lc_clip_apply_kernel  coefficient + scaling of the gradient + the EMA state update, all from device scalars
// No host synchronisation: the reference's `self.start and self.max_norm <= 0` test is evaluated on the device.
#include "lc_common.h"
#include "lc_kernels.h"
#include "lc_map.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    double a[1] = {v};
    wave_allreduce<1>(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a[0];
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>  // element type of the gradient (lc_map.h)
__global__ __launch_bounds__(kThreads) void lc_sqnorm_kernel(const ClipParams p) {
    __shared__ double red[4];
    __shared__ bool last;
    const long long tid = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long long n4 = p.vec ? p.n >> 2 : 0;
    const T* x = static_cast<const T*>(p.x);
    // four requests in flight per thread: raw words in, no branch between the requests (one behind the end re-reads the last group and
    // adds nothing), conversions at the use -- see lc_dense_lse.h for what each of the three is for
    constexpr int kAhead = 4;
    for (long long i = tid; i < n4; i += kAhead * stride) {
        typename MapRaw4<T>::type r[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) r[u] = map_raw_load4(x + 4 * (i + u * stride < n4 ? i + u * stride : n4 - 1));
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            float4 v = map_raw_cvt4<T>(r[u]);
            if (!(i + u * stride < n4)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            a0 = fmaf(v.x, v.x, a0); a1 = fmaf(v.y, v.y, a1); a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
        }
    }
    for (long long i = (n4 << 2) + tid; i < p.n; i += stride) a0 = fmaf((float)x[i], (float)x[i], a0);
    const double part = block_sum_d(((double)a0 + (double)a1) + ((double)a2 + (double)a3), red);
    if (threadIdx.x == 0) {
        // written through the caches and acknowledged before the block counts itself in, read around them by the last block: no
        // agent-scope fence (an L2 write-back per block; lc_common.h: xcd_store)
        xcd_store(p.partials + blockIdx.x, part);
        xcd_stores_done();
        last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    double s = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads) s += xcd_load(p.partials + i);
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) {
        *p.sq = (p.accumulate ? *p.sq : 0.f) + (float)s;
        if (p.state_snapshot) *p.state_snapshot = *p.state_in;  // lets lc_clip_apply update the state in place (hipGraph-safe)
        xcd_store(p.ticket, 0u);  // ready for the next call on this stream
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void lc_clip_apply_kernel(const ClipParams p) {
    const float norm = sqrtf(*p.sq);
    const float state = *p.state_in;
    const bool fresh = state <= 0.f;  // grad.py:20: no running maximum yet
    const float limit = fresh ? p.initial_max_norm : state;
    const float coef = fminf(limit / (norm + 1e-6f), 1.f);  // grad.py:76-80
    const long long tid = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
    const long long n4 = p.vec ? p.n >> 2 : 0;
    const T* x = static_cast<const T*>(p.x);
    T* out = static_cast<T*>(p.out);
    for (long long i = tid; i < n4; i += stride) {
        const float4 v = map_load4(x + 4 * i);
        const float o[4] = {v.x * coef, v.y * coef, v.z * coef, v.w * coef};
        map_store<4>(out + 4 * i, o);
    }
    for (long long i = (n4 << 2) + tid; i < p.n; i += stride) out[i] = map_round<T>((float)x[i] * coef);
    if (tid == 0 && p.state_out) {
        // grad.py:22 / :26-27 in the reference's fp32 operation order
        *p.state_out = fresh ? norm * p.scale : state * p.keep + p.gain * fminf(norm, state * p.scale);
        if (p.norm_out) *p.norm_out = norm;
    }
}

int grid_for(long long n) {
    long long g = (n + (long long)kThreads * 8 - 1) / ((long long)kThreads * 8);
    return (int)(g < 1 ? 1 : (g > kClipMaxBlocks ? kClipMaxBlocks : g));
}

// The sum of squares ends with one counted arrival per workgroup on ONE word (11-13 ns each, served one after the other): a gradient of a
// few MB is reduced by fewer, longer workgroups -- 32 elements per thread (two rounds of four four-element requests) before the grid grows:
// 1 M fp16 elements (zlmo's weight logits): 512 -> 128 workgroups, 9.7 -> 6.x us
int grid_for_sqnorm(long long n) {
    long long g = (n + (long long)kThreads * 32 - 1) / ((long long)kThreads * 32);
    return (int)(g < 1 ? 1 : (g > kClipMaxBlocks ? kClipMaxBlocks : g));
}

}  // namespace

int launch_sqnorm(const ClipParams& p, hipStream_t stream) {
    LC_MAP_DISPATCH(p.dtype, hipLaunchKernelGGL(lc_sqnorm_kernel<T>, dim3(grid_for_sqnorm(p.n)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_clip_apply(const ClipParams& p, hipStream_t stream) {
    LC_MAP_DISPATCH(p.dtype, hipLaunchKernelGGL(lc_clip_apply_kernel<T>, dim3(grid_for(p.n)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
