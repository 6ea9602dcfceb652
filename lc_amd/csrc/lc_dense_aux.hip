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
#include <cstdint>

#include "lc_common.h"
#include "lc_kernels.h"
#include "lc_map.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }  // torch.sigmoid's formula
// the same from the two hardware instructions v_exp_f32 / v_rcp_f32 (<= 1 ulp each): the code loss's backward pass, where ~25 instructions
// of expf + IEEE division per logit were a third of the kernel
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
template <int NWAVES = kThreads / 64>
__device__ __forceinline__ void block_sum3(double (&v)[3], double (*red)[3]) {
    wave_allreduce<3>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 3; ++k) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w += 4) s += (red[w][k] + red[w + 1][k]) + (red[w + 2][k] + red[w + 3][k]);  // fixed order
        v[k] = s;
    }
}

// V consecutive pixels of one map row per thread and request (V = 4: 16-byte loads / stores; HW % 4 == 0 and aligned maps)
template <int V>
struct Px {
    float v[V];
};
// E: float (targets, masks) or the element type of the network's maps (lc_map.h: fp32 / fp16 / bf16, one 16- or 8-byte access)
template <int V, typename E>
__device__ __forceinline__ Px<V> ld(const E* q) {
    Px<V> o;
    map_load<V>(q, o.v);
    return o;
}
template <int V, typename E>
__device__ __forceinline__ void st(E* q, const Px<V>& o) { map_store<V>(q, o.v); }
template <int V>
__device__ __forceinline__ Px<V> mask_at(const DenseAuxParams& p, size_t i) {
    Px<V> o;
    if (p.msk_noc_u8) {
        if constexpr (V == 4) {
            const uchar4 t = *reinterpret_cast<const uchar4*>(p.msk_noc_u8 + i);
            o.v[0] = t.x ? 1.f : 0.f; o.v[1] = t.y ? 1.f : 0.f; o.v[2] = t.z ? 1.f : 0.f; o.v[3] = t.w ? 1.f : 0.f;
        } else {
            o.v[0] = p.msk_noc_u8[i] ? 1.f : 0.f;
        }
        return o;
    }
    return ld<V>(p.msk_noc_f32 + i);
}

// Forward terms with the hardware exp / log (see bce_logits below: ~1e-7 absolute on terms of a mean compared at 1e-6)
template <int SEG>
__device__ __forceinline__ float seg_term(float x, float t) {
    if constexpr (SEG == 0) return (1.f - t) * x - (fminf(x, 0.f) - __logf(1.f + __expf(-fabsf(x))));
    else return fabsf(1.f / (1.f + __expf(-x)) - t);
}

// SEG: 0 BCE-with-logits, 1 Loss_seg_L1 (a template argument: twelve run-time branches per request kept the scheduler from moving anything)
template <int V, int SEG, typename T>
__global__ __launch_bounds__(kThreads) void lc_dense_aux_fwd_kernel(const DenseAuxParams p) {
    __shared__ double red[4][3];
    __shared__ bool last;
    const T* const xyz_map = static_cast<const T*>(p.xyz);
    const T* const seg_map = static_cast<const T*>(p.seg_logits);
    const T* const w_map = static_cast<const T*>(p.wlogits);
    const unsigned xyz_bs = (unsigned)p.xyz_bs, seg_bs = (unsigned)p.seg_bs, wl_bs = (unsigned)p.wl_bs;  // sample strides of the inputs (channel slices in place)
    // 32-bit index arithmetic throughout (the entry points refuse maps of 2^31 elements or more): a 64-bit division per request costs
    // more instructions than the request's arithmetic
    const unsigned HW = (unsigned)p.HW, n = (unsigned)p.B * HW, nv = n / V;
    const bool has_xyz = p.xyz != nullptr, has_w = p.wlogits != nullptr, mask_bytes = p.msk_noc_u8 != nullptr;
    double acc[3] = {0, 0, 0};
    for (unsigned j = blockIdx.x * kThreads + threadIdx.x; j < nv; j += gridDim.x * kThreads) {
        const unsigned i = j * V, b = i / HW, px = i - b * HW;
        // every request of the iteration first (up to eleven of 16 bytes), the arithmetic after them: as written before -- a map's
        // loads next to its arithmetic -- an iteration was FOUR dependent round trips (mask, coordinates, mask logits, weight logits)
        const Px<V> t = ld<V>(p.msk_vis + i), z = ld<V>(seg_map + (b * seg_bs + px));
        Px<V> x[3], g[3], mf, w0, w1;
        unsigned mb = 0;
        if (has_xyz) {
            if (mask_bytes) {
                if constexpr (V == 4) mb = *reinterpret_cast<const unsigned*>(p.msk_noc_u8 + i);
                else mb = p.msk_noc_u8[i];
            } else {
                mf = ld<V>(p.msk_noc_f32 + i);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned e = (b * 3 + c) * HW + px;
                x[c] = ld<V>(xyz_map + (b * xyz_bs + c * HW + px));
                g[c] = ld<V>(p.noc_tgt + e);
            }
        }
        if (has_w) {
            const unsigned e = b * wl_bs + px;
            w0 = ld<V>(w_map + e);
            w1 = ld<V>(w_map + e + HW);
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;  // the V pixels' terms in fp32, one fp64 add per request and loss
        if (has_xyz) {
            Px<V> m;
#pragma unroll
            for (int v = 0; v < V; ++v) m.v[v] = mask_bytes ? (((mb >> (8 * v)) & 0xffu) ? 1.f : 0.f) : mf.v[v];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) s0 += fabsf(x[c].v[v] * m.v[v] - g[c].v[v]);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) s1 += seg_term<SEG>(z.v[v], t.v[v]);
        if (has_w) {
#pragma unroll
            for (int v = 0; v < V; ++v) s2 += seg_term<SEG>(w0.v[v], t.v[v]) + seg_term<SEG>(w1.v[v], t.v[v]);
        }
        acc[0] += (double)s0; acc[1] += (double)s1; acc[2] += (double)s2;
    }
    block_sum3(acc, red);
    if (threadIdx.x == 0) {
        // the block's partials are written through the caches and acknowledged before the block counts itself in; the last block
        // reads all of them around the caches: no agent-scope fence (an L2 write-back per block, ~10 us for 512 blocks: lc_common.h)
        for (int k = 0; k < 3; ++k) xcd_store(p.partials + 3 * blockIdx.x + k, acc[k]);
        xcd_stores_done();
        last = arrive_is_last(p.ticket, blockIdx.x, gridDim.x);
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
    }
    arrival_reset(p.ticket, threadIdx.x);  // ready for the next call on this stream
}

template <int SEG>
__device__ __forceinline__ float seg_term_grad(float x, float t) {
    const float s = 1.f / (1.f + __expf(-x));  // torch.sigmoid's formula on the hardware exp (|error| < 1e-7 on a value in [0, 1])
    if constexpr (SEG == 0) return s - t;
    const float d = s - t;
    return (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (s * (1.f - s));  // sign(s - t) * sigmoid'(x)
}

// Loads first, arithmetic and stores after them (as in the forward kernel): written map by map, a request was EIGHT dependent round
// trips -- the compiler cannot move a map's loads above the previous map's stores (the pointers may alias)
template <int V, int SEG, typename T>
__global__ __launch_bounds__(kThreads) void lc_dense_aux_bwd_kernel(const DenseAuxParams p) {
    const unsigned HW = (unsigned)p.HW, n = (unsigned)p.B * HW, nv = n / V;
    const T* const xyz_map = static_cast<const T*>(p.xyz);
    const T* const seg_map = static_cast<const T*>(p.seg_logits);
    const T* const w_map = static_cast<const T*>(p.wlogits);
    T* const d_xyz = static_cast<T*>(p.d_xyz);  // gradients in the maps' own type
    T* const d_seg = static_cast<T*>(p.d_seg);
    T* const d_wl = static_cast<T*>(p.d_wlogits);
    const unsigned xyz_bs = (unsigned)p.xyz_bs, seg_bs = (unsigned)p.seg_bs, wl_bs = (unsigned)p.wl_bs;  // inputs only: the gradient maps are dense
    const float g0 = (p.g_noc && p.d_xyz) ? *p.g_noc / (3.f * (float)n) : 0.f;
    const float g1 = p.g_seg ? *p.g_seg / (float)n : 0.f;
    const float g2 = (p.g_wseg && p.d_wlogits) ? *p.g_wseg / (2.f * (float)n) : 0.f;
    const bool do_xyz = p.d_xyz != nullptr, do_seg = p.d_seg != nullptr, do_w = p.d_wlogits != nullptr, mask_bytes = p.msk_noc_u8 != nullptr;
    for (unsigned j = blockIdx.x * kThreads + threadIdx.x; j < nv; j += gridDim.x * kThreads) {
        const unsigned i = j * V, b = i / HW, px = i - b * HW;
        const unsigned ex = b * 3 * HW + px, ew = b * 2 * HW + px;
        const Px<V> t = ld<V>(p.msk_vis + i);
        Px<V> x[3], g[3], mf, z, w[2];
        unsigned mb = 0;
        if (do_xyz) {
            if (mask_bytes) {
                if constexpr (V == 4) mb = *reinterpret_cast<const unsigned*>(p.msk_noc_u8 + i);
                else mb = p.msk_noc_u8[i];
            } else {
                mf = ld<V>(p.msk_noc_f32 + i);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = ld<V>(xyz_map + (b * xyz_bs + px + c * HW));
                g[c] = ld<V>(p.noc_tgt + ex + c * HW);
            }
        }
        if (do_seg) z = ld<V>(seg_map + (b * seg_bs + px));
        if (do_w) {
            w[0] = ld<V>(w_map + (b * wl_bs + px));
            w[1] = ld<V>(w_map + (b * wl_bs + px + HW));
        }
        if (do_xyz) {
            Px<V> m;
#pragma unroll
            for (int v = 0; v < V; ++v) m.v[v] = mask_bytes ? (((mb >> (8 * v)) & 0xffu) ? 1.f : 0.f) : mf.v[v];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Px<V> o;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float d = x[c].v[v] * m.v[v] - g[c].v[v];
                    o.v[v] = (d > 0.f ? g0 : (d < 0.f ? -g0 : 0.f)) * m.v[v];  // torch.sign(0) = 0
                }
                st<V>(d_xyz + ex + c * HW, o);
            }
        }
        if (do_seg) {
            Px<V> o;
#pragma unroll
            for (int v = 0; v < V; ++v) o.v[v] = g1 * seg_term_grad<SEG>(z.v[v], t.v[v]);
            st<V>(d_seg + i, o);
        }
        if (do_w) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Px<V> o;
#pragma unroll
                for (int v = 0; v < V; ++v) o.v[v] = g2 * seg_term_grad<SEG>(w[c].v[v], t.v[v]);
                st<V>(d_wl + ew + c * HW, o);
            }
        }
    }
}

// ---- Loss_xyz_bin (losses.py:196-216): per-bit weighted BCE on the binary surface-code logits with an EMA histogram of per-bit
// Hamming errors.  The reference walks the (B,C,H,W) logits ~12 times (compare, xor, and, two integer reductions, mask multiply,
// log-sigmoid, BCE, mean, ...); here one pass: grid (C, chunks), a workgroup reduces its share of ONE code channel -- Hamming errors
// inside the hard visibility mask, the BCE sum, and (channel 0) the mask's population -- partials in a fixed order, and the last
// workgroup to arrive finishes: histogram EMA in place, soft histogram, softmax bit weights, the loss.  Backward: one element-wise pass.
#ifndef LC_BIN_FWD_AHEAD
#define LC_BIN_FWD_AHEAD 1
#endif
constexpr int kBinChunks = 32;   // at most that many workgroups per code channel (their partials are added in chunk order)
constexpr int kBinThreads = 1024;  // of 1024 threads each: every workgroup ends with ONE counted arrival on the one counter

// (1 - t) z - log_sigmoid(z), log_sigmoid(z) = min(z, 0) - log1p(exp(-|z|)); the hardware exp / log (v_exp_f32, v_log_f32) are good
// to ~1e-7 absolute on log1p(e), e in (0, 1] -- terms of a MEAN of order 0.1-1 that is compared at 1e-6
// (`__logf` is NOT the bare instruction under hipcc: it adds a denormal-range rescue and a two-term ln 2 product, ~9 instructions; the
// argument here is in (1, 2], so log2 from v_log_f32 times ln 2 is all that is needed -- 28 of the kernel's 125 VALU instructions per request)
__device__ __forceinline__ float softplus_neg_abs(float z) {  // log(1 + exp(-|z|))
    return 0.693147180559945309f * __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(z)));
}
__device__ __forceinline__ float bce_logits(float z, float t) { return (1.f - t) * z - (fminf(z, 0.f) - softplus_neg_abs(z)); }

// The end of Loss_xyz_bin.forward from the per-bit error counts (losses.py:205-213), shared by the one-launch kernel (its last workgroup)
// and by the weights-in kernel of the sharded form, so that both give the same bits: histogram EMA in place, soft histogram, softmax
// over the C <= 128 bits, the weighted sum of the per-bit BCE means parked in `ws`.  Called by every thread of the workgroup
// (blockDim >= C); hamm / ws hold C valid entries and are visible (a barrier has been passed).
__device__ __forceinline__ void bin_loss_finish(const BinLossParams& p, const long long* hamm, long long vis_total, float h_old, float* zs, float* ws, int tid) {
    __shared__ float es[kBinMaxChannels];
    if (tid < p.C) {
        // losses.py:205-208 in the reference's fp32 operation order (integer tensors divide as float32)
        const float hist = (float)hamm[tid] / (float)(vis_total + 1);
        float h = h_old;  // p.histogram[tid], requested by the caller before its other loads (a memory round trip at the very end of the launch otherwise)
        h = h * (1.f - p.momentum);
        h = h + hist * p.momentum;
        p.histogram[tid] = h;
        zs[tid] = fminf(h, 0.51f - h) * 3.f;
    }
    __syncthreads();
    // one thread per bit for the exponentials, sums in index order (every thread forms them for itself from LDS): the one-thread
    // version cost ~3 us of serial expf at the very end of the launch
    float m = -INFINITY;
    for (int ch = 0; ch < p.C; ++ch) m = fmaxf(m, zs[ch]);
    if (tid < p.C) es[tid] = expf(zs[tid] - m);
    __syncthreads();
    float sum = 0.f;
    for (int ch = 0; ch < p.C; ++ch) sum += es[ch];
    if (tid < p.C) {
        const float w = es[tid] / sum;
        p.bin_weights[tid] = w;
        ws[tid] *= w;
    }
    __syncthreads();
    if (tid == 0) {
        float loss = 0.f;
        for (int ch = 0; ch < p.C; ++ch) loss += ws[ch];
        *p.loss = loss;
    }
}

template <typename T>
__global__ __launch_bounds__(kBinThreads) void lc_xyz_bin_loss_fwd_kernel(const BinLossParams p) {
    const T* const logits = static_cast<const T*>(p.logits);
    const T* const vis_logits = static_cast<const T*>(p.msk_vis_logits);
    const size_t lg_bs = (size_t)p.logits_bs, vis_bs = (size_t)p.vis_bs;  // sample strides of the inputs
    __shared__ double red[kBinThreads / 64][3];
    __shared__ bool last;
    __shared__ float zs[kBinMaxChannels], ws[kBinMaxChannels];
    const int chunks = p.chunks, c = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const unsigned n = (unsigned)p.B * (unsigned)p.HW;  // 32-bit index arithmetic (see lc_dense_aux_fwd_kernel)
    // Hamming errors and visible pixels (channel 0 only) are counted in integers; the BCE terms of the four pixels of a request are
    // added in fp32 and that sum goes into a double (one conversion + one fp64 add per request: the kernel is VALU-bound)
    int n_err = 0, n_vis = 0, e4 = 0, v4 = 0;  // e4 / v4 / bce4: the counts and the sum of one request's four pixels
    double bce_sum = 0;
    float bce4 = 0.f;
    auto one = [&](float x, bool t, bool vis) {
        e4 += (vis && ((x > 0.f) != t)) ? 1 : 0;
        bce4 += bce_logits(vis ? x : 0.f * x, t ? 1.f : 0.f);  // logits * msk_hard (keeps a NaN / inf logit visible like the product)
        v4 += (c == 0 && vis) ? 1 : 0;
    };
    if (p.vec) {  // HW % 4 == 0, 16-byte aligned maps: four pixels per thread and request
        const unsigned hw4 = (unsigned)p.HW >> 2;
        constexpr int kAhead = LC_BIN_FWD_AHEAD;  // requests in flight per thread: the loop is a latency chain otherwise (one round trip per iteration)
        // (prefetching the NEXT four requests while these are evaluated was tried: 94 VGPRs -> one 1024-thread workgroup per compute
        // unit instead of two, 33 -> 39 us at B=64 128x128)
        // The channel's pixels are walked PLANE by plane: a unit = kBinThreads requests (one per thread) of one sample's plane, units dealt
        // to the channel's workgroups round robin (balanced to one unit: with units of 4 x 1024 requests a third of the workgroups had
        // twice the work of the others at zlmo's shape, 28 us against 23).  Inside a unit every address is a uniform base + the thread's
        // offset -- the flat walk over (sample, pixel) paid two integer divisions per request (a quarter of the kernel's VALU
        // instructions).  kAhead units are in flight per thread: all their requests are issued before the first is used -- raw words in, no
        // branch between the requests (a unit behind the workgroup's last re-reads that one and is not counted), conversions at the use.
        const unsigned slabs = (hw4 + kBinThreads - 1) / kBinThreads, units = (unsigned)p.B * slabs;
        for (unsigned u0 = (unsigned)chunk; u0 < units; u0 += kAhead * (unsigned)chunks) {
            typename MapRaw4<T>::type xr[kAhead], vr[kAhead];
            uchar4 t[kAhead];
            bool live[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const unsigned unit = min(u0 + u * (unsigned)chunks, units - 1);  // uniform
                const unsigned b = unit / slabs, q0 = (unit - b * slabs) * kBinThreads + threadIdx.x, q = min(q0, hw4 - 1);
                live[u] = u0 + u * (unsigned)chunks < units && q0 < hw4;
                xr[u] = map_raw_load4(logits + ((size_t)b * lg_bs + (size_t)c * (unsigned)p.HW) + 4 * (size_t)q);
                t[u] = (reinterpret_cast<const uchar4*>(p.gt_bits) + ((size_t)b * (unsigned)p.C + (unsigned)c) * hw4)[q];
                vr[u] = map_raw_load4(vis_logits + (size_t)b * vis_bs + 4 * (size_t)q);
            }
            // (evaluated without a branch either: with `if (...) break` here the compiler sinks each request's loads into its guarded
            // block and the group is one request in flight again)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const float4 x = map_raw_cvt4<T>(xr[u]), v = map_raw_cvt4<T>(vr[u]);
                one(x.x, t[u].x != 0, v.x > 0.f); one(x.y, t[u].y != 0, v.y > 0.f);
                one(x.z, t[u].z != 0, v.z > 0.f); one(x.w, t[u].w != 0, v.w > 0.f);
                if (live[u]) {
                    n_err += e4; n_vis += v4;
                    bce_sum += (double)bce4;
                }
                e4 = 0; v4 = 0; bce4 = 0.f;
            }
        }
    } else {
        for (unsigned i = (unsigned)chunk * kBinThreads + threadIdx.x; i < n; i += (unsigned)chunks * kBinThreads) {
            const unsigned b = i / (unsigned)p.HW, px = i - b * (unsigned)p.HW;
            const unsigned e = (b * (unsigned)p.C + (unsigned)c) * (unsigned)p.HW + px;
            one((float)logits[(size_t)b * lg_bs + (size_t)c * (unsigned)p.HW + px], p.gt_bits[e] != 0, (float)vis_logits[(size_t)b * vis_bs + px] > 0.f);
            n_err += e4; n_vis += v4;
            bce_sum += (double)bce4;
            e4 = 0; v4 = 0; bce4 = 0.f;
        }
    }
    double acc[3] = {(double)n_err, bce_sum, (double)n_vis};
    block_sum3<kBinThreads / 64>(acc, red);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) xcd_store(p.partials + 3 * blockIdx.x + k, acc[k]);
        xcd_stores_done();
        last = arrive_is_last(p.ticket, blockIdx.x, gridDim.x);
    }
    __syncthreads();
    if (!last) return;
    // The last workgroup: every partial is fetched around the caches ONCE, many requests in flight per thread (a thread adding its
    // channel's chunks one dependent load after the other spent ~0.7 us per load), staged in LDS and added in chunk order.
    const int tid = threadIdx.x;
    // for bin_loss_finish: the histogram's entry in flight together with the partials (the counts-out form has no histogram: any readable word)
    float h_old = (p.counts_out ? reinterpret_cast<const float*>(p.partials) : p.histogram)[min(tid, p.C - 1)];
    asm volatile("" : "+v"(h_old));
    __shared__ double stage[32][2][kBinChunks];
    __shared__ double vis_part[kBinChunks];
    __shared__ long long hamm_s[kBinMaxChannels];
    __shared__ long long vis_s;
    if (tid < chunks) vis_part[tid] = xcd_load(p.partials + 3 * tid + 2);  // channel 0's chunks
    for (int g0 = 0; g0 < p.C; g0 += 32) {
        for (int v = tid; v < 32 * 2 * chunks; v += kBinThreads) {
            const int ch = v / (2 * chunks), r = v % (2 * chunks), which = r / chunks, k = r % chunks;
            if (g0 + ch < p.C) stage[ch][which][k] = xcd_load(p.partials + 3 * ((g0 + ch) * chunks + k) + which);
        }
        __syncthreads();
        if (tid < 32 && g0 + tid < p.C) {
            const int ch = g0 + tid;
            double hamm = 0, bce = 0, vis_total = 0;
            for (int k = 0; k < chunks; ++k) { hamm += stage[tid][0][k]; bce += stage[tid][1][k]; vis_total += vis_part[k]; }
            hamm_s[ch] = (long long)hamm;  // integers below 2^53: exact
            if (ch == 0) vis_s = (long long)vis_total;
            ws[ch] = (float)(bce / (double)n);  // loss_raw.mean([0, 2, 3]) of this bit, parked until the weights exist
        }
        __syncthreads();
    }
    __syncthreads();
    if (p.counts_out) {
        // counts-out form (the batch is sharded over ranks): this rank's C error counts + its visible-pixel count leave as 64-bit integers
        // for the all-reduce, the rank's per-bit BCE means wait in `bce_mean` for lc_xyz_bin_loss_finish_kernel
        if (tid < p.C) { p.counts_out[tid] = hamm_s[tid]; p.bce_mean[tid] = ws[tid]; }
        if (tid == 0) p.counts_out[p.C] = vis_s;
    } else {
        bin_loss_finish(p, hamm_s, vis_s, h_old, zs, ws, tid);
    }
    arrival_reset(p.ticket, tid);
}

// weights-in half of the sharded form: the all-reduced counts -> histogram EMA, bit weights, this rank's loss (one small workgroup)
__global__ __launch_bounds__(kBinMaxChannels) void lc_xyz_bin_loss_finish_kernel(const BinLossParams p) {
    __shared__ float zs[kBinMaxChannels], ws[kBinMaxChannels];
    __shared__ long long hamm_s[kBinMaxChannels];
    const int tid = threadIdx.x;
    const float h_old = p.histogram[min(tid, p.C - 1)];
    const long long vis_total = p.counts_in[p.C];
    if (tid < p.C) { hamm_s[tid] = p.counts_in[tid]; ws[tid] = p.bce_mean[tid]; }
    __syncthreads();
    bin_loss_finish(p, hamm_s, vis_total, h_old, zs, ws, tid);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void lc_xyz_bin_loss_bwd_kernel(const BinLossParams p) {
    const T* const logits = static_cast<const T*>(p.logits);
    const T* const vis_logits = static_cast<const T*>(p.msk_vis_logits);
    T* const d_logits = static_cast<T*>(p.d_logits);
    const size_t lg_bs = (size_t)p.logits_bs, vis_bs = (size_t)p.vis_bs;
    const unsigned n = (unsigned)p.B * (unsigned)p.C * (unsigned)p.HW, C = (unsigned)p.C;
    const float g = *p.g_loss / (float)((unsigned)p.B * (unsigned)p.HW);
    // d/dx BCE(x * m, t) = (sigmoid(x m) - t) m
    auto one = [&](float x, bool t, bool vis, float w) { return vis ? g * w * (sigmoid_fast(x) - (t ? 1.f : 0.f)) : 0.f; };  // (the same bits as the plane-shaped kernel below)
    if (p.vec) {
        const unsigned hw4 = (unsigned)p.HW >> 2;
        for (unsigned e = blockIdx.x * kThreads + threadIdx.x; e < (n >> 2); e += gridDim.x * kThreads) {
            const unsigned bc = e / hw4, q = e - bc * hw4, b = bc / C;
            const float w = p.bin_weights[bc - b * C];
            const float4 x = map_load4(logits + ((size_t)b * lg_bs + 4 * (size_t)((bc - b * C) * hw4 + q)));
            const uchar4 t = reinterpret_cast<const uchar4*>(p.gt_bits)[e];
            const float4 v = map_load4(vis_logits + ((size_t)b * vis_bs + 4 * (size_t)q));
            const float o[4] = {one(x.x, t.x != 0, v.x > 0.f, w), one(x.y, t.y != 0, v.y > 0.f, w), one(x.z, t.z != 0, v.z > 0.f, w),
                                one(x.w, t.w != 0, v.w > 0.f, w)};
            map_store<4>(d_logits + 4 * (size_t)e, o);
        }
        return;
    }
    for (unsigned e = blockIdx.x * kThreads + threadIdx.x; e < n; e += gridDim.x * kThreads) {
        const unsigned bc = e / (unsigned)p.HW, px = e - bc * (unsigned)p.HW, b = bc / C;
        d_logits[e] = map_round<T>(one((float)logits[(size_t)b * lg_bs + (size_t)(bc - b * C) * (unsigned)p.HW + px], p.gt_bits[e] != 0,
                             (float)vis_logits[(size_t)b * vis_bs + px] > 0.f, p.bin_weights[bc - b * C]));
    }
}

// The same backward pass shaped by the (sample, code bit) PLANES of the maps: grid (plane chunks, C, B), so a workgroup's plane base
// addresses, its bit weight and the cotangent are uniform (scalar registers, no integer division per request -- the flat kernel above spends
// two on every four pixels), and each thread issues 16-byte requests -- E = 4 fp32 or 8 sixteen-bit pixels -- two per thread, all loads of
// both in flight before the first is used.  sigmoid(x) = v_rcp_f32(1 + v_exp_f32(-x log2 e)): two hardware instructions of <= 1 ulp each
// against ~25 for expf + an IEEE division (__frcp_rn still expands to v_div_scale / v_div_fmas / v_div_fixup: the builtins are used directly) -- a gradient entry moves by ~1e-7 of itself (the tests compare at 2e-6 of the largest entry).
// Element-wise: every map type evaluates the same expression per element, so a 16-bit map still gives the fp32 result on its up-cast
// values, rounded once to the map's type.
#ifndef LC_BIN_BWD_UNROLL
#define LC_BIN_BWD_UNROLL 2
#endif
constexpr int kBwdPlaneUnroll = LC_BIN_BWD_UNROLL;
template <int E, typename T>
struct PlaneChunk {  // one 16-byte request of logits and visibility logits + its E ground-truth bytes, as loaded (converted at the use)
    uint4 x, v;
    unsigned t[E / 4];
};
template <int E, typename T>
__device__ __forceinline__ void plane_load(const T* x, const unsigned char* t, const T* v, size_t at, PlaneChunk<E, T>& o) {
    static_assert(E * sizeof(T) == 16, "16-byte requests");
    // (`at` counts elements and is a multiple of E: indexing typed 16-byte / E-byte pointers keeps the alignment visible to the compiler --
    // with byte arithmetic one of the 16-byte requests came out as four unaligned pieces)
    const size_t chunk = at / E;
    o.x = static_cast<const uint4*>(__builtin_assume_aligned(x, 16))[chunk];
    o.v = static_cast<const uint4*>(__builtin_assume_aligned(v, 16))[chunk];
    if constexpr (E == 4) {
        o.t[0] = reinterpret_cast<const unsigned*>(t)[chunk];
    } else {
        const uint2 b = reinterpret_cast<const uint2*>(t)[chunk];
        o.t[0] = b.x; o.t[1] = b.y;
    }
}
template <int E, typename T>
__device__ __forceinline__ void plane_unpack(const PlaneChunk<E, T>& q, float (&x)[E], float (&v)[E], bool (&t)[E]) {
    const unsigned wx[4] = {q.x.x, q.x.y, q.x.z, q.x.w}, wv[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if constexpr (sizeof(T) == 4) {
            x[k] = __builtin_bit_cast(float, wx[k]);
            v[k] = __builtin_bit_cast(float, wv[k]);
        } else {  // (extracted by shifts: through a memcpy into T[8] one of the 16-byte requests was split into four narrow loads)
            x[k] = (float)__builtin_bit_cast(T, (unsigned short)(wx[k / 2] >> (16 * (k % 2))));
            v[k] = (float)__builtin_bit_cast(T, (unsigned short)(wv[k / 2] >> (16 * (k % 2))));
        }
        t[k] = ((q.t[k / 4] >> (8 * (k % 4))) & 0xffu) != 0;
    }
}
template <int E, typename T>
__device__ __forceinline__ void plane_store(T* d, size_t at, const float (&o)[E]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<map_v4f_t*>(d + at) = map_v4f_t{o[0], o[1], o[2], o[3]};
    } else {
        T h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = map_round<T>(o[k]);
        uint4 r;
        __builtin_memcpy(&r, h, 16);
        *reinterpret_cast<uint4*>(d + at) = r;
    }
}
template <int E, typename T>
__global__ __launch_bounds__(kThreads) void lc_xyz_bin_loss_bwd_plane_kernel(const BinLossParams p) {
    const unsigned c = blockIdx.y, b = blockIdx.z, HW = (unsigned)p.HW, chunks = HW / E;
    const T* const x = static_cast<const T*>(p.logits) + ((size_t)b * (size_t)p.logits_bs + (size_t)c * HW);
    const T* const v = static_cast<const T*>(p.msk_vis_logits) + (size_t)b * (size_t)p.vis_bs;
    const size_t plane = ((size_t)b * (unsigned)p.C + c) * HW;
    const unsigned char* const t = p.gt_bits + plane;
    T* const d = static_cast<T*>(p.d_logits) + plane;
    const float g = *p.g_loss / (float)((unsigned)p.B * HW), w = p.bin_weights[c];
    // d/dx BCE(x * m, t) = (sigmoid(x m) - t) m, in the flat kernel's operation order
    PlaneChunk<E, T> q[kBwdPlaneUnroll];
    const unsigned first = blockIdx.x * (kBwdPlaneUnroll * kThreads) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < kBwdPlaneUnroll; ++u)  // no branch between the requests: one behind the plane's end re-reads the last chunk and is dropped below
        plane_load<E, T>(x, t, v, (size_t)min(first + u * kThreads, chunks - 1) * E, q[u]);
#pragma unroll
    for (int u = 0; u < kBwdPlaneUnroll; ++u) {
        if (first + u * kThreads >= chunks) break;
        float xs[E], vs[E], o[E];
        bool ts[E];
        plane_unpack<E, T>(q[u], xs, vs, ts);
#pragma unroll
        for (int k = 0; k < E; ++k) o[k] = vs[k] > 0.f ? g * w * (sigmoid_fast(xs[k]) - (ts[k] ? 1.f : 0.f)) : 0.f;
        plane_store<E, T>(d, (size_t)(first + u * kThreads) * E, o);
    }
}

int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    return (int)(g < 1 ? 1 : (g > kDenseAuxMaxBlocks ? kDenseAuxMaxBlocks : g));
}

}  // namespace

static DenseAuxParams aux_with_dense_strides(DenseAuxParams p) {  // batch strides left at 0 mean dense batches
    if (!p.xyz_bs) p.xyz_bs = 3ll * p.HW;
    if (!p.seg_bs) p.seg_bs = p.HW;
    if (!p.wl_bs) p.wl_bs = 2ll * p.HW;
    return p;
}
static BinLossParams bin_with_dense_strides(BinLossParams p) {
    if (!p.logits_bs) p.logits_bs = (long long)p.C * p.HW;
    if (!p.vis_bs) p.vis_bs = p.HW;
    return p;
}

static bool aux_vec(const DenseAuxParams& p) {
    const auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    const uintptr_t m = 4 * (uintptr_t)map_elem_bytes(p.map_dtype);  // four elements of a network map
    return p.HW % 4 == 0 && ((p.xyz_bs | p.seg_bs | p.wl_bs) & 3) == 0 && al(p.xyz, m) && al(p.noc_tgt, 16) && al(p.seg_logits, m) && al(p.msk_vis, 16) && al(p.wlogits, m) &&
           al(p.msk_noc_f32, 16) && al(p.msk_noc_u8, 4) && al(p.d_xyz, m) && al(p.d_seg, m) && al(p.d_wlogits, m);
}

int launch_dense_aux_fwd(const DenseAuxParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    const DenseAuxParams p = aux_with_dense_strides(p_in);
    // one request of four pixels per thread up to four blocks per compute unit (the arrivals are counted on sharded words: lc_common.h)
    const long long n = (long long)p.B * p.HW;
    const int grid = (int)std::min<long long>(kDenseAuxMaxBlocks, std::max<long long>(1, (n + 4 * kThreads - 1) / (4 * kThreads)));
    const bool vec = aux_vec(p);
    LC_MAP_DISPATCH(p.map_dtype,
                    if (p.seg_type == 0) {
                        if (vec) hipLaunchKernelGGL((lc_dense_aux_fwd_kernel<4, 0, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                        else hipLaunchKernelGGL((lc_dense_aux_fwd_kernel<1, 0, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                    } else {
                        if (vec) hipLaunchKernelGGL((lc_dense_aux_fwd_kernel<4, 1, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                        else hipLaunchKernelGGL((lc_dense_aux_fwd_kernel<1, 1, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                    });
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_aux_bwd(const DenseAuxParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    const DenseAuxParams p = aux_with_dense_strides(p_in);
    const bool vec = aux_vec(p);
    const int grid = grid_for(((long long)p.B * p.HW) / (vec ? 4 : 1));
    LC_MAP_DISPATCH(p.map_dtype,
                    if (p.seg_type == 0) {
                        if (vec) hipLaunchKernelGGL((lc_dense_aux_bwd_kernel<4, 0, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                        else hipLaunchKernelGGL((lc_dense_aux_bwd_kernel<1, 0, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                    } else {
                        if (vec) hipLaunchKernelGGL((lc_dense_aux_bwd_kernel<4, 1, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                        else hipLaunchKernelGGL((lc_dense_aux_bwd_kernel<1, 1, T>), dim3(grid), dim3(kThreads), 0, stream, p);
                    });
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_xyz_bin_loss_fwd(const BinLossParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.C <= 0) return 0;
    if (p.C > kBinMaxChannels) return 3;
    // 1024-thread workgroups, a thread at about four requests of four pixels (all in flight at once), at most kBinChunks workgroups
    // per code bit (B=64 128x128: 16 -> 32 chunks 38 -> 33 us; B=32 64x64: eight -> four requests 16.4 -> 12.5 us)
    BinLossParams q = bin_with_dense_strides(p);
    // the vectorised walk deals units of 1024 requests of ONE sample's plane; the element-wise walk strides over (sample, pixel)
    const long long per_plane = ((long long)p.HW / 4 + kBinThreads - 1) / kBinThreads;
    const long long units = p.vec ? (long long)p.B * per_plane : (((long long)p.B * p.HW + 3) / 4 + 4 * kBinThreads - 1) / (4 * kBinThreads);
    q.chunks = (int)std::min<long long>(kBinChunks, std::max<long long>(1, units));
    q.chunks = std::min(q.chunks, std::max(1, 512 / q.C));  // one round of workgroups: two of 1024 threads fit a compute unit
    LC_MAP_DISPATCH(q.map_dtype, hipLaunchKernelGGL(lc_xyz_bin_loss_fwd_kernel<T>, dim3(q.C * q.chunks), dim3(kBinThreads), 0, stream, q));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_xyz_bin_loss_finish(const BinLossParams& p, hipStream_t stream) {
    if (p.C <= 0) return 0;
    if (p.C > kBinMaxChannels) return 3;
    hipLaunchKernelGGL(lc_xyz_bin_loss_finish_kernel, dim3(1), dim3(kBinMaxChannels), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_xyz_bin_loss_bwd(const BinLossParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0 || p_in.C <= 0) return 0;
    const BinLossParams p = bin_with_dense_strides(p_in);
    const long long n = (long long)p.B * p.C * p.HW;
    // planes of 16-byte requests (the training shapes: 64x64 / 128x128 maps, contiguous or channel-sliced heads), else the flat kernel
    const int E = 16 / map_elem_bytes(p.map_dtype);
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool planes = p.vec && p.HW % E == 0 && p.logits_bs % E == 0 && p.vis_bs % E == 0 && al16(p.logits) && al16(p.msk_vis_logits) && al16(p.d_logits) &&
                        (reinterpret_cast<uintptr_t>(p.gt_bits) & (E - 1)) == 0 && p.B <= 65535;
    if (planes) {
        const dim3 grid((p.HW / E + kBwdPlaneUnroll * kThreads - 1) / (kBwdPlaneUnroll * kThreads), p.C, p.B);
        if (p.map_dtype == kMapF32) hipLaunchKernelGGL((lc_xyz_bin_loss_bwd_plane_kernel<4, float>), grid, dim3(kThreads), 0, stream, p);
        else if (p.map_dtype == kMapF16) hipLaunchKernelGGL((lc_xyz_bin_loss_bwd_plane_kernel<8, _Float16>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((lc_xyz_bin_loss_bwd_plane_kernel<8, __bf16>), grid, dim3(kThreads), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    const int grid = (int)std::min<long long>(2048, std::max<long long>(1, (n + 4 * kThreads - 1) / (4 * kThreads)));
    LC_MAP_DISPATCH(p.map_dtype, hipLaunchKernelGGL(lc_xyz_bin_loss_bwd_kernel<T>, dim3(grid), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
