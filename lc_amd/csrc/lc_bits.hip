// ZebraPose-style binary surface codes -> normalised object coordinates.
//
// Replaces floatbits.py:130-160 (mod_logits2float_with_gt_bb_scripted: differentiable MSB-error decode used in training,
// ~15 elementwise ops + gather/scatter over (B,H,W,3,bits)) and floatbits.py:194-223 (mod_logits2float_bb: Gray-code
// decode used at inference), including the /(max/2)-1 normalisation of floatbits.py:108-118,162-180 and the strided
// sub-sampling of losses.py:163-184.  Works directly on the network layout (B,C,H,W), C = n0+n1+n2 code bits, one thread
// per four consecutive pixels of a row (float4 / uchar4 channel reads, float4 stores; one pixel per thread when the row
// length or the sub-sampling stride does not allow it); nothing is permuted or materialised.
#include <algorithm>
#include <cstdint>

#include "lc_common.h"
#include "lc_kernels.h"
#include "lc_map.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

struct AxisDecode {
    float val;     // decoded value in [0, 2^n - 1]
    int idx;       // channel (within the axis) that carries the gradient, or -1
    float dval;    // d val / d logit[idx]
};

// floatbits.py:130-160 for one axis of one pixel, fed one code bit (channel) at a time, MSB first
struct AxisState {
    float out_val = 0.f, correct = 0.f, x_idx = 0.f, sgn_idx = 1.f;
    int idx = 0;
    bool found = false, prev = false;

    __device__ __forceinline__ void push(int k, int n, float logit, bool b, int black_factor) {
        float sgn = (k >= 1 && prev) ? -1.f : 1.f;   // logits_msk[1:] = -1 where the previous gt bit is set
        if (k < 2) sgn *= (float)black_factor;       // logits_msk[0:2] *= black_factor
        const float x = logit * sgn;
        const bool pred = x > 0.f;
        const float w = (float)(1 << (n - 1 - k));
        out_val += pred ? w : 0.f;
        const bool err = (pred != b) || (k == n - 1);
        if (err && !found) {  // first erroneous bit (torch.argmax returns the first maximal index)
            found = true;
            idx = k; x_idx = x; sgn_idx = sgn;
        } else {
            correct += b ? w : 0.f;  // gt bits with the MSB-error bit cleared
        }
        prev = b;
    }
    __device__ __forceinline__ AxisDecode finish(int n, bool in_msk) const {
        AxisDecode o;
        const float w = (float)(1 << (n - 1 - idx));
        const float s = 1.f / (1.f + __expf(-x_idx));
        if (in_msk) {
            o.val = correct + s * w;
            o.idx = idx;
            o.dval = w * s * (1.f - s) * sgn_idx;
        } else {
            o.val = out_val;
            o.idx = -1;
            o.dval = 0.f;
        }
        return o;
    }
};

// V consecutive pixels of one row per thread: one 16-byte (fp32 logits) / 8-byte (fp16, bf16 logits: lc_map.h) / uchar4 channel read when V == 4
template <int V, typename T>
__device__ __forceinline__ void load_px(const T* lg, float (&v)[V]) { map_load<V>(lg, v); }
template <int V>
__device__ __forceinline__ void load_px(const unsigned char* g, bool (&v)[V]) {
    if constexpr (V == 4) {
        const uchar4 t = *reinterpret_cast<const uchar4*>(g);
        v[0] = t.x != 0; v[1] = t.y != 0; v[2] = t.z != 0; v[3] = t.w != 0;
    } else {
        v[0] = g[0] != 0;
    }
}
template <int V, typename T>
__device__ __forceinline__ void store_px(T* o, const float (&v)[V]) {
    map_store<V>(o, v, V == 4 && LC_NT_GRAD_STORES);  // d_logits: written once, read by the next kernel of the backward pass
}

// The channels of an axis are requested kChanBatch at a time, every request of a batch before the first use: walked one channel at a time
// (load, use, load, ...) a wavefront had ONE request in flight, and the decode of 16-bit logits -- half the bytes -- was hardly faster than
// fp32 (17.0 vs 19.3 us at 64 x 21 x 128 x 128: a chain of 21 memory round trips per thread, not bandwidth).  The kernels request the first
// batch of all three axes up front.
#ifndef LC_BITS_GT_PIPELINE
#define LC_BITS_GT_PIPELINE 1
#endif
#ifndef LC_BITS_GT_BATCH
#define LC_BITS_GT_BATCH 1
#endif
#ifndef LC_BITS_FWD_WIDE
#define LC_BITS_FWD_WIDE 1  // A/B switch: 0 = the generic forward kernel for strided subsets too
#endif
#ifndef LC_BITS_BWD_TILES
#define LC_BITS_BWD_TILES 1  // A/B switch: 0 = the flat backward kernel for strided subsets too
#endif
constexpr int kGrayBatch = 8, kGtBatch = LC_BITS_GT_BATCH;  // channels per batch: inference decode / training decode (which also holds the ground-truth bits and the axis state)
template <int V, int kChanBatch>
struct ChanBatch {
    float x[kChanBatch][V];
    unsigned t[kChanBatch];  // the V ground-truth bits of the channel as loaded (one byte each): unpacked where they are used -- a compare next
                             // to the load is a wait for that load
    __device__ __forceinline__ bool bit(int j, int v) const { return ((t[j] >> (8 * v)) & 0xffu) != 0; }
};
template <int V, typename T, int kChanBatch>
__device__ __forceinline__ void load_channels(const T* lg, const unsigned char* gt, size_t stride, int k0, int n, ChanBatch<V, kChanBatch>& q) {
    // branch-free: a slot behind the axis' last channel re-requests that channel (a cache hit) instead of being skipped -- with a branch per
    // slot every request sat in a basic block of its own behind an s_waitcnt vmcnt(0), i.e. one request in flight again
#pragma unroll
    for (int j = 0; j < kChanBatch; ++j) {
        const size_t k = (size_t)min(k0 + j, n - 1);
        load_px<V>(lg + k * stride, q.x[j]);
        if (gt) {
            if constexpr (V == 4) q.t[j] = *reinterpret_cast<const unsigned*>(gt + k * stride);
            else q.t[j] = gt[k * stride];
        }
    }
}

// Eight channels of one axis of ONE pixel, logits and ground-truth bits (both required here), with no branch between the requests: a slot
// behind the axis' last channel re-requests that channel.  (load_channels' `if (gt)` is a run-time test: it splits the requests into basic
// blocks with a wait in each.)
template <typename T>
__device__ __forceinline__ void load_axis8(const T* lg, const unsigned char* gt, size_t stride, int k0, int n, float (&x)[8], unsigned char (&t)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t k = (size_t)min(k0 + j, n - 1);
        x[j] = (float)lg[k * stride];
        t[j] = gt[k * stride];
    }
}

// first: channels 0 .. kChanBatch-1 of the axis, already requested
template <int V, typename T>
__device__ __forceinline__ void decode_axis_gt(const T* lg, const unsigned char* gt, size_t stride, int n, int black_factor,
                                               AxisState (&st)[V], const ChanBatch<V, kGtBatch>& first) {
    constexpr int kChanBatch = kGtBatch;
    auto walk = [&](const ChanBatch<V, kGtBatch>& q, int k0) {
#pragma unroll
        for (int j = 0; j < kChanBatch; ++j) {
            if (k0 + j < n) {
#pragma unroll
                for (int v = 0; v < V; ++v) st[v].push(k0 + j, n, q.x[j][v], q.bit(j, v), black_factor);
            }
        }
    };
#if LC_BITS_GT_PIPELINE
    // one batch ahead: the next batch is requested before the current one is walked (the last request repeats the axis' last channel)
    ChanBatch<V, kGtBatch> cur = first;
    for (int k0 = 0; k0 < n; k0 += kChanBatch) {
        ChanBatch<V, kGtBatch> nxt;
        load_channels<V>(lg, gt, stride, k0 + kChanBatch, n, nxt);
        walk(cur, k0);
        cur = nxt;
    }
#else
    walk(first, 0);
    for (int k0 = kChanBatch; k0 < n; k0 += kChanBatch) {  // more than kChanBatch code bits on the axis
        ChanBatch<V, kGtBatch> more;
        load_channels<V>(lg, gt, stride, k0, n, more);
        walk(more, k0);
    }
#endif
}

// floatbits.py:194-223 for one axis of V pixels
template <int V, typename T>
__device__ __forceinline__ void decode_gray(const T* lg, size_t stride, int n, bool black, float (&out)[V], const ChanBatch<V, kGrayBatch>& first) {
    constexpr int kChanBatch = kGrayBatch;
    unsigned code[V];
    float last[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { code[v] = 0; last[v] = 0.f; }
    auto walk = [&](const ChanBatch<V, kGrayBatch>& q, int k0) {
#pragma unroll
        for (int j = 0; j < kChanBatch; ++j) {
            if (k0 + j < n) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    last[v] = q.x[j][v];
                    bool bit = last[v] > 0.f;
                    if (black && k0 + j < 2) bit = !bit;
                    code[v] = (code[v] << 1) | (bit ? 1u : 0u);
                }
            }
        }
    };
    walk(first, 0);
    for (int k0 = kChanBatch; k0 < n; k0 += kChanBatch) {  // more than kChanBatch code bits on the axis
        ChanBatch<V, kGrayBatch> more;
        load_channels<V>(lg, static_cast<const unsigned char*>(nullptr), stride, k0, n, more);
        walk(more, k0);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        unsigned g = code[v];  // inverse Gray code (the reference's lut[src] = dst with src = dst ^ (dst >> 1))
        for (unsigned sh = 1; sh < 32; sh <<= 1) g ^= g >> sh;
        const float lsb_factor = (g & 2u) ? -1.f : 1.f;
        out[v] = (float)(g & ~1u) + 1.f / (1.f + __expf(-last[v] * lsb_factor));
    }
}

// forward of the training decode.  V == 4 needs sample == 1 (the pixel subset is the whole map), W % 4 == 0
// normalised coordinates -> object coordinates of sample b (nn_out_to_xyz): x = (n * s - t) @ R, R = T[:3,:3], t = T[:3,3]; and the
// cotangent's way back, g_n[i] = s[i] * sum_j g_x[j] R[i][j].  Identity where the pointers are null.
__device__ const float kOutMapIdentity[3 + 16] = {1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};  // scale | 4x4 transform
struct OutMap {
    float s[3], t[3], R[3][3];
    bool scale, xform;
    // The 15 values are read UNCONDITIONALLY through a pointer that falls back to an identity table: written as `scale ? load : 1.f` per
    // entry, every load sat in a branch of its own behind an s_waitcnt vmcnt(0) -- eight dependent memory round trips at the head of every
    // workgroup (seen in the ISA of the backward kernels; ~1 us each).
    __device__ OutMap(const BitsParams& p, int b) : scale(p.out_scale != nullptr), xform(p.out_xform != nullptr) {
        const float* const sc = scale ? p.out_scale + 3 * (size_t)b : kOutMapIdentity;
        const float* const xf = xform ? p.out_xform + 16 * (size_t)b : kOutMapIdentity + 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            s[i] = sc[i];
            t[i] = xf[4 * i + 3];
#pragma unroll
            for (int j = 0; j < 3; ++j) R[i][j] = xf[4 * i + j];
        }
    }
    __device__ __forceinline__ void apply(float (&v)[3]) const {
        if (!scale && !xform) return;
        const float w0 = v[0] * s[0] - t[0], w1 = v[1] * s[1] - t[1], w2 = v[2] * s[2] - t[2];
        if (!xform) { v[0] = w0; v[1] = w1; v[2] = w2; return; }
        for (int j = 0; j < 3; ++j) v[j] = w0 * R[0][j] + w1 * R[1][j] + w2 * R[2][j];
    }
    __device__ __forceinline__ float pull(const float* g, int i) const {  // g: the three cotangents of one point
        if (!xform) return g[i] * s[i];
        return s[i] * (g[0] * R[i][0] + g[1] * R[i][1] + g[2] * R[i][2]);
    }
};

template <int V, typename T>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_fwd_kernel(const BitsParams p) {
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.N / V;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const size_t i0 = i * V;
        const int b = (int)(i0 / p.N), n = (int)(i0 % p.N);
        const int y = p.top + (n / Wn) * p.sample, x = p.left + (n % Wn) * p.sample;
        const size_t px = (size_t)y * p.W + x;
        bool in_msk[V];
        if (p.gt_msk) {
            load_px<V>(p.gt_msk + (size_t)b * HW + px, in_msk);
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) in_msk[v] = true;
        }
        float res[V][3];
        ChanBatch<V, kGtBatch> q[3];
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {  // every axis' first batch of channels in flight together
            load_channels<V>(static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs + (size_t)c0 * HW + px, p.gt_bits + ((size_t)b * p.C + c0) * HW + px,
                             HW, 0, p.bits[a], q[a]);
            c0 += p.bits[a];
        }
        c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            const size_t base = ((size_t)b * p.C + c0) * HW + px, lbase = (size_t)b * p.logits_bs + (size_t)c0 * HW + px;
            AxisState st[V];
            decode_axis_gt<V>(static_cast<const T*>(p.logits) + lbase, p.gt_bits + base, HW, nb, p.black_factor, st, q[a]);
#pragma unroll
            for (int v = 0; v < V; ++v) res[v][a] = st[v].finish(nb, in_msk[v]).val / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
            c0 += nb;
        }
        const OutMap om(p, b);  // the V pixels of a request belong to one sample (N % V == 0 on this path)
#pragma unroll
        for (int v = 0; v < V; ++v) om.apply(res[v]);
        float* o = p.out + i0 * 3;
        if constexpr (V == 4) {  // 12 contiguous floats
            *reinterpret_cast<float4*>(o) = make_float4(res[0][0], res[0][1], res[0][2], res[1][0]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(res[1][1], res[1][2], res[2][0], res[2][1]);
            *reinterpret_cast<float4*>(o + 8) = make_float4(res[2][2], res[3][0], res[3][1], res[3][2]);
        } else {
            o[0] = res[0][0]; o[1] = res[0][1]; o[2] = res[0][2];
        }
    }
}

// forward on a STRIDED subset (training: losses.py:163-184 with dense_sample 2 or 3), one thread per sampled pixel.  The generic kernel above
// walks an axis one channel ahead of its use: ~8 dependent memory round trips per axis, and a strided subset has too few threads (zlmo: 59 k
// = less than one wavefront per SIMD) for anything to hide them -- 12 us for 4.5 MB.  Here every request of the pixel -- the object mask
// and the first eight bits of all three axes, logits and ground truth: up to 49 loads -- is issued before the first is used (one round
// trip; axes of more than eight bits take further rounds).  Same AxisState sequence, same OutMap: the same bits.
template <typename T>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_fwd_wide_kernel(const BitsParams p) {
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const int b = blockIdx.y, n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= p.N) return;
    const int ry = n / Wn;
    const size_t px = (size_t)(p.top + ry * p.sample) * p.W + p.left + (n - ry * Wn) * p.sample;
    const T* const logits = static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs + px;
    const unsigned char* const gt = p.gt_bits + (size_t)b * p.C * HW + px;
    const unsigned char m = (p.gt_msk ? p.gt_msk + (size_t)b * HW : p.gt_bits + (size_t)b * p.C * HW)[px];
    float xs[3][8];
    unsigned char ts[3][8];
    int c0 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        load_axis8(logits + (size_t)c0 * HW, gt + (size_t)c0 * HW, HW, 0, p.bits[a], xs[a], ts[a]);
        c0 += p.bits[a];
    }
    const OutMap om(p, b);
    const bool in_msk = m != 0 || !p.gt_msk;
    float res[3];
    c0 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int nb = p.bits[a];
        AxisState st;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            if (jj < nb) st.push(jj, nb, xs[a][jj], ts[a][jj] != 0, p.black_factor);
        for (int k0 = 8; k0 < nb; k0 += 8) {
            load_axis8(logits + (size_t)c0 * HW, gt + (size_t)c0 * HW, HW, k0, nb, xs[a], ts[a]);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                if (k0 + jj < nb) st.push(k0 + jj, nb, xs[a][jj], ts[a][jj] != 0, p.black_factor);
        }
        res[a] = st.finish(nb, in_msk).val / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
        c0 += nb;
    }
    om.apply(res);
    float* o = p.out + ((size_t)b * p.N + n) * 3;
    o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
}

// backward: one thread per V pixels of the FULL map: writes every channel (zeros where no gradient arrives)
template <int V, typename T>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_bwd_kernel(const BitsParams p) {
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * HW / V;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const size_t i0 = i * V;
        const int b = (int)(i0 / HW);
        const size_t px = i0 % HW;
        const int y = (int)(px / p.W), x0 = (int)(px % p.W);
        const int dy = y - p.top;
        const bool row_hit = dy >= 0 && (dy % p.sample) == 0;
        bool live[V], any = false;
        size_t n[V];
#pragma unroll
        for (int v = 0; v < V; ++v) live[v] = false;
        if (row_hit) {
            if (p.gt_msk) {
                load_px<V>(p.gt_msk + (size_t)b * HW + px, live);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) live[v] = true;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int dx = x0 + v - p.left;
            live[v] = row_hit && live[v] && dx >= 0 && (dx % p.sample) == 0;
            n[v] = live[v] ? (size_t)(dy / p.sample) * Wn + dx / p.sample : 0;
            any = any || live[v];
        }
        const OutMap om(p, b);
        ChanBatch<V, kGtBatch> q[3];
        int c0 = 0;
        if (any) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {  // every axis' first batch of channels in flight together
                load_channels<V>(static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs + (size_t)c0 * HW + px,
                                 p.gt_bits + ((size_t)b * p.C + c0) * HW + px, HW, 0, p.bits[a], q[a]);
                c0 += p.bits[a];
            }
        }
        c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            const size_t base = ((size_t)b * p.C + c0) * HW + px, lbase = (size_t)b * p.logits_bs + (size_t)c0 * HW + px;
            int idx[V];
            float g[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { idx[v] = -1; g[v] = 0.f; }
            if (any) {
                AxisState st[V];
                decode_axis_gt<V>(static_cast<const T*>(p.logits) + lbase, p.gt_bits + base, HW, nb, p.black_factor, st, q[a]);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (live[v]) {
                        const AxisDecode d = st[v].finish(nb, true);
                        idx[v] = d.idx;
                        g[v] = om.pull(p.g_out + ((size_t)b * p.N + n[v]) * 3, a) * d.dval / ((float)((1 << nb) - 1) * 0.5f);
                    }
                }
            }
            for (int k = 0; k < nb; ++k) {
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) o[v] = (k == idx[v]) ? g[v] : 0.f;
                store_px<V>(static_cast<T*>(p.d_logits) + base + k * HW, o);
            }
            c0 += nb;
        }
    }
}

// The same backward pass for the STRIDED subsets of training (losses.py:163-184 with dense_sample 2 or 3: one pixel in four or nine carries
// a gradient, in ONE of the bits of each axis; everything else of the (B,C,H,W) gradient map is zero).  The flat kernel above gives a thread
// four pixels of ALL channels: at zlmo's shape (32 x 21 x 128 x 128, stride 3) that is 131 k threads -- two wavefronts per SIMD -- each
// writing 21 eight-byte pieces, 1.1 TB/s.  Here a workgroup owns up to kTileRows rows of one sample:
//   1. decode: one thread per (sampled pixel, axis) of the tile -- every channel of the axis requested at once (V = 1, batches of eight) --
//      leaves the bit that carries the gradient and its value in LDS (-1: no gradient: outside the object mask);
//   2. write: each thread owns one 16-byte piece (E = 4 fp32 / 8 sixteen-bit pixels) of the tile's rows and walks the channels, every
//      piece written ONCE with the sampled pixels' values merged in from LDS -- no partial writes, no read-modify-write.
// Per pixel the arithmetic is the flat kernel's (AxisState, OutMap::pull, the same expression for the value): the same bits.
constexpr int kTileRows = 4;       // at most; fewer where more would give a thread two items of the decode (zlmo: 3 rows = one sampled row of 43 pixels x 3 axes)
template <int E, typename T>
__device__ __forceinline__ void tile_store(T* at, const float (&o)[E]) {
    // non-temporal: written once, read by the next kernel of the backward pass (plain stores measured +1.7 us at zlmo's shape)
    if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(map_v4f_t{o[0], o[1], o[2], o[3]}, reinterpret_cast<map_v4f_t*>(at));
    } else {
        T h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = map_round<T>(o[k]);
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        v4u_t r;
        __builtin_memcpy(&r, h, 16);
        __builtin_nontemporal_store(r, reinterpret_cast<v4u_t*>(at));
    }
}
// Thread layout: blockDim = (W / E pieces of a row, kThreads / that): threadIdx.y = row of the tile + R x (thread group); SAMPLE and R are
// template arguments -- with run-time values the index arithmetic (a dozen integer divisions per thread, two per pixel of a piece) was most
// of the kernel's 4.9 M wavefront instructions.
template <int E, typename T, int SAMPLE, int R>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_bwd_tile_kernel(const BitsParams p) {
    static_assert(E * sizeof(T) == 16, "16-byte pieces");
    constexpr int kLiveRows = (R + SAMPLE - 1) / SAMPLE;  // sampled rows a tile can hold
    __shared__ int s_idx[kThreads];
    __shared__ float s_g[kThreads];
    const int b = blockIdx.y, y0 = blockIdx.x * R, tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int Wn = (p.W - p.left + SAMPLE - 1) / SAMPLE;
    const size_t HW = (size_t)p.H * p.W;
    // sampled rows of the tile: r_first = the first sampled row index whose y >= y0
    const int r_first = y0 <= p.top ? 0 : (y0 - p.top + SAMPLE - 1) / SAMPLE;
    const int y_end = min(y0 + R, p.H);
    const int r_end = y_end <= p.top ? 0 : (y_end - p.top + SAMPLE - 1) / SAMPLE;  // sampled rows below y_end
    const int rows_live = max(0, r_end - r_first), live = rows_live * Wn * 3;
    // this thread's piece of the tile plane, and the thread group that shares the channels with it
    const int groups = p.tile_groups, grp = (int)threadIdx.y / R, yy = (int)threadIdx.y - grp * R, x0 = (int)threadIdx.x * E, y = y0 + yy;
    const bool writer = grp < groups && y < p.H;
    const int dy = y - p.top;
    const bool row_hit = dy >= 0 && dy % SAMPLE == 0;
    T* const d = static_cast<T*>(p.d_logits) + (size_t)b * p.C * HW + (size_t)y * p.W + x0;
    // rows without a sampled pixel are zero in every channel: written first (after the decode instead: the same time, measured)
    if (writer && !row_hit) {
        const float z[E] = {};
        for (int c = grp; c < p.C; c += groups) tile_store<E, T>(d + (size_t)c * HW, z);
    }
    const T* const logits = static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs;
    const unsigned char* const gt = p.gt_bits + (size_t)b * p.C * HW;
    const OutMap om(p, b);
    if (tid < live) {  // (the launcher sizes the tile so that live <= kThreads: every item has a thread of its own)
        const int i = tid, pxi = i / 3, a = i - 3 * pxi;  // (pixel-major; an axis-major item order measured slower)
        int rr = 0, j = pxi;
        if constexpr (kLiveRows > 1) { rr = pxi / Wn; j = pxi - rr * Wn; }
        const int r = r_first + rr;
        const size_t px = (size_t)(p.top + r * SAMPLE) * p.W + p.left + j * SAMPLE;
        // (p.bits[a] with a per-lane `a` would be a VECTOR load from the kernel-argument segment -- host memory: 25 us of PCIe round trips
        // in the first version of this kernel; the three counts come in by scalar loads and are selected)
        const int nb0 = p.bits[0], nb1 = p.bits[1], nb2 = p.bits[2];
        const int c0 = a == 0 ? 0 : (a == 1 ? nb0 : nb0 + nb1), nb = a == 0 ? nb0 : (a == 1 ? nb1 : nb2);
        // everything the item reads is requested before anything is used -- the object mask (no mask: any readable byte, overridden below),
        // the axis' first eight bits (logit + ground truth), the point's cotangent: one memory round trip, not three
        const unsigned char m = (p.gt_msk ? p.gt_msk + (size_t)b * HW : gt)[px];
        float xs[8];
        unsigned char ts[8];
        load_axis8(logits + (size_t)c0 * HW + px, gt + (size_t)c0 * HW + px, HW, 0, nb, xs, ts);
        const float* const gp = p.g_out + ((size_t)b * p.N + (size_t)r * Wn + j) * 3;
        const float g3[3] = {gp[0], gp[1], gp[2]};
        AxisState st;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            if (jj < nb) st.push(jj, nb, xs[jj], ts[jj] != 0, p.black_factor);
        for (int k0 = 8; k0 < nb; k0 += 8) {  // axes of more than eight bits
            load_axis8(logits + (size_t)c0 * HW + px, gt + (size_t)c0 * HW + px, HW, k0, nb, xs, ts);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                if (k0 + jj < nb) st.push(k0 + jj, nb, xs[jj], ts[jj] != 0, p.black_factor);
        }
        const AxisDecode dd = st.finish(nb, true);
        const bool in_msk = m != 0 || !p.gt_msk;
        s_idx[i] = in_msk ? dd.idx : -1;
        const float pulled = a == 0 ? om.pull(g3, 0) : (a == 1 ? om.pull(g3, 1) : om.pull(g3, 2));  // (a constant index each: registers, not a per-thread LDS copy of the map)
        s_g[i] = in_msk ? pulled * dd.dval / ((float)((1 << nb) - 1) * 0.5f) : 0.f;
    }
    __syncthreads();
    if (!writer || !row_hit) return;
    // the rows that hold sampled pixels: the piece's sampled pixels first move from LDS to registers (bit index and value per axis), then
    // every channel of the piece is ONE 16-byte store with those values merged in
    int pi[E][3];
    float pg[E][3];
    const int row_slot = (dy / SAMPLE - r_first) * Wn;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int dx = x0 + k - p.left;
        const bool hit = dx >= 0 && dx % SAMPLE == 0;
        const int slot = hit ? (row_slot + dx / SAMPLE) * 3 : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pi[k][a] = hit ? s_idx[slot + a] : -1;
            pg[k][a] = hit ? s_g[slot + a] : 0.f;
        }
    }
    int c0 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int nb = p.bits[a];
        // channels c0 .. c0 + nb - 1 of this axis; this group's: c = grp (mod groups)
        int kbit = (grp - c0 % groups + groups) % groups;
        for (; kbit < nb; kbit += groups) {
            float o[E];
#pragma unroll
            for (int k = 0; k < E; ++k) o[k] = pi[k][a] == kbit ? pg[k][a] : 0.f;
            tile_store<E, T>(d + (size_t)(c0 + kbit) * HW, o);
        }
        c0 += nb;
    }
}

template <int E, typename T>
int launch_tile_bwd(const BitsParams& pt, int sample, int rows, dim3 grid, dim3 block, hipStream_t stream) {
#define LC_TILE_CASE(S, RR) \
    if (sample == S && rows == RR) { hipLaunchKernelGGL((lc_bits_decode_gt_bwd_tile_kernel<E, T, S, RR>), grid, block, 0, stream, pt); return 1; }
    LC_TILE_CASE(2, 1) LC_TILE_CASE(2, 2) LC_TILE_CASE(2, 3) LC_TILE_CASE(2, 4)
    LC_TILE_CASE(3, 1) LC_TILE_CASE(3, 2) LC_TILE_CASE(3, 3) LC_TILE_CASE(3, 4)
#undef LC_TILE_CASE
    return 0;
}

template <int V, typename T>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_kernel(const BitsParams p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * HW / V;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const size_t i0 = i * V;
        const int b = (int)(i0 / HW);
        const size_t px = i0 % HW;
        float res[V][3];
        ChanBatch<V, kGrayBatch> q[3];
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {  // every axis' first batch of channels in flight together
            load_channels<V>(static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs + (size_t)c0 * HW + px, static_cast<const unsigned char*>(nullptr), HW,
                             0, p.bits[a], q[a]);
            c0 += p.bits[a];
        }
        c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            float val[V];
            decode_gray<V>(static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs + (size_t)c0 * HW + px, HW, nb, p.black_factor < 0, val, q[a]);
#pragma unroll
            for (int v = 0; v < V; ++v) res[v][a] = val[v] / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
            c0 += nb;
        }
        const OutMap om(p, b);
#pragma unroll
        for (int v = 0; v < V; ++v) om.apply(res[v]);
        if (p.out_planar) {  // (B,3,H,W): one coalesced store per coordinate plane
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) o[v] = res[v][a];
                store_px<V>(p.out + ((size_t)b * 3 + a) * HW + px, o);
            }
            continue;
        }
        float* o = p.out + i0 * 3;
        if constexpr (V == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(res[0][0], res[0][1], res[0][2], res[1][0]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(res[1][1], res[1][2], res[2][0], res[2][1]);
            *reinterpret_cast<float4*>(o + 8) = make_float4(res[2][2], res[3][0], res[3][1], res[3][2]);
        } else {
            o[0] = res[0][0]; o[1] = res[0][1]; o[2] = res[0][2];
        }
    }
}

// Inference decode of the SELECTED pixels only.  At test time the point selection keeps a fraction of the candidates (zlmo: 80 % of the visible
// fifth of 128x128 = ~3300 of 16 384 per object), and only those need model coordinates: decoding the whole map first (lc_bits_decode_kernel: 19 us
// per 64 objects, 101 MB read, 12.6 MB written and read back) did five times the work.  One thread per selected entry, its pixel found from the
// selection's source index; the same per-pixel arithmetic as the whole-map decode (decode_gray<1>, OutMap) -> the same floats.
template <typename T>
__global__ __launch_bounds__(kThreads) void lc_bits_decode_rows_kernel(const BitsParams p) {
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const int b = blockIdx.y;
    const int n = min(p.rows_counts[b], p.rows_N);
    const OutMap om(p, b);
    const T* const logits = static_cast<const T*>(p.logits) + (size_t)b * p.logits_bs;
    for (int k = blockIdx.x * kThreads + threadIdx.x; k < n; k += gridDim.x * kThreads) {
        const int idx = p.rows_index[(size_t)b * p.rows_N + k];
        const int r = idx / Wn;
        const size_t px = (size_t)(p.top + r * p.sample) * p.W + p.left + (idx - r * Wn) * p.sample;
        float res[3];
        ChanBatch<1, kGrayBatch> q[3];
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            load_channels<1>(logits + (size_t)c0 * HW + px, static_cast<const unsigned char*>(nullptr), HW, 0, p.bits[a], q[a]);
            c0 += p.bits[a];
        }
        c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            float val[1];
            decode_gray<1>(logits + (size_t)c0 * HW + px, HW, nb, p.black_factor < 0, val, q[a]);
            res[a] = val[0] / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
            c0 += nb;
        }
        om.apply(res);
        float* o = p.out + ((size_t)b * p.rows_N + k) * 3;
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
    }
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
bool aligned4(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; }

BitsParams with_dense_stride(BitsParams p) {  // a batch stride left at 0 means a dense batch
    if (!p.logits_bs) p.logits_bs = (long long)p.C * p.H * p.W;
    return p;
}

int grid_for(size_t total) {
    size_t g = (total + kThreads - 1) / kThreads;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // cap + grid-stride (cdna_hip_programming.md Guideline 11)
}

}  // namespace

int launch_bits_decode_gt_fwd(const BitsParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0 || p_in.N <= 0) return 0;
    const BitsParams p = with_dense_stride(p_in);
    const bool vec = (p.logits_bs % 4) == 0 && p.sample == 1 && p.top == 0 && p.left == 0 && p.W % 4 == 0 && map_aligned4(p.logits, p.map_dtype) && aligned16(p.out) && aligned4(p.gt_bits) && aligned4(p.gt_msk);
    if (LC_BITS_FWD_WIDE && !vec && p.B <= 65535) {  // strided subsets (and odd shapes): every request of a pixel in flight at once
        LC_MAP_DISPATCH(p.map_dtype, hipLaunchKernelGGL(lc_bits_decode_gt_fwd_wide_kernel<T>, dim3((p.N + kThreads - 1) / kThreads, p.B), dim3(kThreads), 0, stream, p));
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    LC_MAP_DISPATCH(p.map_dtype,
                    if (vec) hipLaunchKernelGGL((lc_bits_decode_gt_fwd_kernel<4, T>), dim3(grid_for((size_t)p.B * p.N / 4)), dim3(kThreads), 0, stream, p);
                    else hipLaunchKernelGGL((lc_bits_decode_gt_fwd_kernel<1, T>), dim3(grid_for((size_t)p.B * p.N)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_bits_decode_gt_bwd(const BitsParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    const BitsParams p = with_dense_stride(p_in);
    const bool vec = (p.logits_bs % 4) == 0 && p.W % 4 == 0 && map_aligned4(p.logits, p.map_dtype) && map_aligned4(p.d_logits, p.map_dtype) && aligned4(p.gt_bits) && aligned4(p.gt_msk);
    // strided subsets (training): tiles of kTileRows rows, 16-byte pieces
    const int E = 16 / map_elem_bytes(p.map_dtype);
    const int Wn = p.sample > 0 ? (p.W - p.left + p.sample - 1) / p.sample : 0;
    // rows per tile: as many as give every thread at most one (sampled pixel, axis) item of the decode and one 16-byte piece of the plane
    const int per_row = p.W % E == 0 ? p.W / E : 0;
    int tile_rows = 0;
    for (int R = kTileRows; R >= 1 && !tile_rows && per_row > 0 && kThreads % per_row == 0; --R)
        if (((R + p.sample - 1) / p.sample) * Wn * 3 <= kThreads && R * per_row <= kThreads) tile_rows = R;
    if (LC_BITS_BWD_TILES && (p.sample == 2 || p.sample == 3) && tile_rows > 0 && aligned16(p.d_logits) && p.B <= 65535) {
        BitsParams pt = p;
        pt.tile_groups = (kThreads / per_row) / tile_rows;
        const dim3 grid((p.H + tile_rows - 1) / tile_rows, p.B), block(per_row, kThreads / per_row);
        int ok = 0;
        if (p.map_dtype == kMapF32) ok = launch_tile_bwd<4, float>(pt, p.sample, tile_rows, grid, block, stream);
        else if (p.map_dtype == kMapF16) ok = launch_tile_bwd<8, _Float16>(pt, p.sample, tile_rows, grid, block, stream);
        else ok = launch_tile_bwd<8, __bf16>(pt, p.sample, tile_rows, grid, block, stream);
        if (ok) return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    LC_MAP_DISPATCH(p.map_dtype,
                    if (vec) hipLaunchKernelGGL((lc_bits_decode_gt_bwd_kernel<4, T>), dim3(grid_for((size_t)p.B * p.H * p.W / 4)), dim3(kThreads), 0, stream, p);
                    else hipLaunchKernelGGL((lc_bits_decode_gt_bwd_kernel<1, T>), dim3(grid_for((size_t)p.B * p.H * p.W)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_bits_decode(const BitsParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0) return 0;
    const BitsParams p = with_dense_stride(p_in);
    const bool vec = (p.logits_bs % 4) == 0 && p.W % 4 == 0 && map_aligned4(p.logits, p.map_dtype) && aligned16(p.out);
    LC_MAP_DISPATCH(p.map_dtype,
                    if (vec) hipLaunchKernelGGL((lc_bits_decode_kernel<4, T>), dim3(grid_for((size_t)p.B * p.H * p.W / 4)), dim3(kThreads), 0, stream, p);
                    else hipLaunchKernelGGL((lc_bits_decode_kernel<1, T>), dim3(grid_for((size_t)p.B * p.H * p.W)), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_bits_decode_rows(const BitsParams& p_in, hipStream_t stream) {
    if (p_in.B <= 0 || p_in.rows_N <= 0) return 0;
    const BitsParams p = with_dense_stride(p_in);
    const int gx = std::max(1, std::min(16, (p.rows_N + kThreads - 1) / kThreads));  // a row's count is usually a fraction of its length: grid-stride over it
    LC_MAP_DISPATCH(p.map_dtype, hipLaunchKernelGGL(lc_bits_decode_rows_kernel<T>, dim3(gx, p.B), dim3(kThreads), 0, stream, p));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
