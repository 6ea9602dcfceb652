// Test-time point selection for the dense heads (second half): per-sample mask / weight-quantile
// selection of the dense correspondences and their COMPACTION into padded per-sample lists + counts, on the device.
//
// Replaces test.py:39-45 (quantile_msk: torch.quantile over the summed weights), test.py:94-106 (the three
// dense_point_select modes, one nonzero() per sample = one host sync per sample, ragged Python lists, np.random padding)
// and the re-batching of those lists in cer_solver.py:67-87.  One workgroup per sample:
//   1. weight_i = inv_std[i,0] + inv_std[i,1]  (x seg_i in mode 2)
//   2. modes 1/2: radix select of the two order statistics around q*(n-1) from the weights staged in LDS, threshold =
//      torch.quantile's linear interpolation between them  (mode 2: q_b = 1 - (1-q) * mean(seg), test.py:102-103)
//   3. keep_i = seg_i | weight_i >= thr | (weight_i >= thr) & seg_i
//   4. order-preserving compaction (wave ballots + prefix over the waves) of pts2d, weights (optionally squared: the
//      inverse covariance the solver takes, test.py:92), pts3d and the source indices; count per sample
//   5. fewer than min_count survivors (and more than min_count candidates): pad with pseudo-random source indices, the
//      role np.random.choice plays in test.py:108-113 (seeded hash instead of the host RNG)
// The outputs feed lc_pnp_ransac_init_f32 / lc_pnp_lm_f32 directly through their `counts` argument.
#include <cfloat>

#include "lc_common.h"
#include "lc_kernels.h"
#include "lc_select_rows.h"

namespace lc {
namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / kWave;

// torch.lerp (aten/src/ATen/native/Lerp.h): the form that is exact at both ends
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
    return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

__global__ __launch_bounds__(kThreads) void lc_dense_select_kernel(const SelectParams p) {
    extern __shared__ float srt[];  // n order-preserving integer keys of the weights (modes 1, 2)
    __shared__ int wave_cnt[kWaves];
    __shared__ int s_seg;
    __shared__ float s_thr;
    __shared__ int hist[256];
    __shared__ int s_sel[2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(p.in_counts ? p.in_counts[b] : p.N, p.N);
    const size_t base = (size_t)b * p.N;
    const float* ws = p.inv_std + base * 2;
    const unsigned char* seg = p.mask ? p.mask + base : nullptr;
    auto weight = [&](int i) {
        const float2 s = *reinterpret_cast<const float2*>(ws + 2 * i);
        return (p.mode == 2 && !seg[i]) ? 0.f : s.x + s.y;  // mode 2: (inv_std * seg).sum(-1)
    };
    if (tid == 0) { s_seg = 0; s_thr = -FLT_MAX; }
    __syncthreads();
    if (p.mode != 0 && n > 0) {
        // torch.quantile needs only the two order statistics around q (n - 1): a most-significant-digit RADIX SELECT over the
        // order-preserving integer image of the weights (4 passes of 8 bits: LDS histogram, one-wave scan) finds the lower one in
        // O(n), one counting pass the upper one -- the bitonic sort it replaces took 78 barrier-separated stages for n = 4096 (44.8 -> ~10 us per launch).  The
        // threshold is formed from the same two floats, so the selected index sets are unchanged bit for bit.
        unsigned* keys = reinterpret_cast<unsigned*>(srt);
        int segc = 0;
        for (int i = tid; i < n; i += kThreads) {
            const unsigned u = __float_as_uint(weight(i));
            keys[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending unsigned order == ascending float order
            if (p.mode == 2 && seg[i]) ++segc;
        }
        if (p.mode == 2) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) segc += __shfl_xor(segc, m, kWave);
            if (lane == 0) atomicAdd(&s_seg, segc);
        }
        __syncthreads();
        float q = p.quantile;
        if (p.mode == 2) q = 1.f - p.one_minus_q * ((float)s_seg / (float)n);  // test.py:102-103, fp32 like the tensor op
        q = fminf(fmaxf(q, 0.f), 1.f);
        const float rank = q * (float)(n - 1);
        const float lo = floorf(rank), hi = ceilf(rank);
        const int klo = (int)lo, khi = min((int)hi, n - 1);
        // rank klo by the radix select; rank khi = klo + 1 is then either the same key (a duplicate: more than klo + 1 keys are <= it)
        // or the smallest key above it -- one counting / min pass instead of four more histogram passes
        unsigned found[2] = {0u, 0u};
        {
            unsigned prefix = 0u, mask = 0u;
            int k = klo;  // 0-based rank among the elements that still match the prefix
            for (int pass = 3; pass >= 0; --pass) {
                const int shift = 8 * pass;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                for (int i = tid; i < n; i += kThreads) {
                    const unsigned key = keys[i];
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
                }
                __syncthreads();
                if (wave == 0) {  // lane l owns bins 4l .. 4l+3: exclusive prefix over the lanes, then the bin holding rank k
                    const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                    const int mine = c0 + c1 + c2 + c3;
                    int incl = mine;
#pragma unroll
                    for (int d = 1; d < kWave; d <<= 1) {
                        const int up = __shfl_up(incl, d, kWave);
                        if (lane >= d) incl += up;
                    }
                    const int excl = incl - mine;
                    if (k >= excl && k < incl) {  // exactly one lane
                        int r = k - excl, bin = 4 * lane;
                        if (r >= c0) { r -= c0; ++bin; if (r >= c1) { r -= c1; ++bin; if (r >= c2) { r -= c2; ++bin; } } }
                        s_sel[0] = bin;
                        s_sel[1] = r;
                    }
                }
                __syncthreads();
                prefix |= (unsigned)s_sel[0] << shift;
                mask |= 255u << shift;
                k = s_sel[1];
            }
            found[0] = found[1] = prefix;
        }
        if (khi != klo) {
            if (tid < 2) hist[tid] = tid == 0 ? 0 : -1;  // hist[0]: #keys <= found[0];  hist[1]: smallest key above it (as unsigned max)
            __syncthreads();
            int le = 0;
            unsigned above = 0xFFFFFFFFu;
            for (int i = tid; i < n; i += kThreads) {
                const unsigned key = keys[i];
                if (key <= found[0]) ++le;
                else above = min(above, key);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                le += __shfl_xor(le, m, kWave);
                above = min(above, (unsigned)__shfl_xor((int)above, m, kWave));
            }
            if (lane == 0) {
                atomicAdd(&hist[0], le);
                atomicMin(reinterpret_cast<unsigned*>(&hist[1]), above);
            }
            __syncthreads();
            if (hist[0] <= khi) found[1] = (unsigned)hist[1];  // no duplicate reaches rank khi: the next distinct key
            __syncthreads();  // hist is reused by nothing below, but keep the read before any later write
        }
        if (tid == 0) {
            auto unkey = [](unsigned kk) { return __uint_as_float((kk & 0x80000000u) ? (kk & 0x7FFFFFFFu) : ~kk); };
            const float vlo = unkey(found[0]), vhi = khi != klo ? unkey(found[1]) : vlo;
            s_thr = torch_lerp(vlo, vhi, rank - lo);
        }
        __syncthreads();
    }
    const float thr = s_thr;
    const RowCopy rows{p.pts2d, p.inv_std, p.pts3d, p.in_index, p.o_pts2d, p.o_w, p.o_pts3d, p.o_index, p.square};
    int running = 0;  // survivors in the chunks before this one (same value in every thread)
    for (int i0 = 0; i0 < n; i0 += kThreads) {
        const int i = i0 + tid;
        bool keep = false;
        if (i < n) {
            if (p.mode == 0) keep = seg[i] != 0;
            else keep = weight(i) >= thr && (p.mode == 1 || seg[i] != 0);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = running, tot = running;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            if (w < wave) off += wave_cnt[w];
            tot += wave_cnt[w];
        }
        if (keep) rows.entry(base, i, off + before);
        running = tot;
        __syncthreads();  // wave_cnt is rewritten by the next chunk
    }
    const int total = rows.pad(base, b, n, running, p.min_count, p.seed);  // test.py:108-113
    if (tid == 0) p.counts[b] = total;
}

}  // namespace

int launch_dense_select(const SelectParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    int P = 1;
    while (P < p.N) P <<= 1;
    const size_t lds = p.mode == 0 ? 0 : (size_t)P * sizeof(float);
    if (lds > 128 * 1024) return 3;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(lc_dense_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 2;
    hipLaunchKernelGGL(lc_dense_select_kernel, dim3(p.B), dim3(kThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
