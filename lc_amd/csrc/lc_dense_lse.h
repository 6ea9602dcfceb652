// Log-sum-exp of a sample's weight logits (the joint softmax of losses.py:355): shared by the dense front end (lc_dense.hip) and the
// test-time kernel that runs front end and point selection in one launch (lc_select.hip).  NT threads of the workgroup take part
// (the same NT, stride and merge order everywhere -> the same float everywhere); the workgroup may be larger.
#pragma once
#include <cfloat>
#include <cstdint>

#include "lc_common.h"
#include "lc_map.h"

namespace lc {

#ifndef LC_LSE_AHEAD
#define LC_LSE_AHEAD 4  // requests in flight per thread in block_lse (A/B: scripts/ubench/graph_inference.py with a -DLC_LSE_AHEAD=1 variant build)
#endif
constexpr int kDenseLseThreads = 512;  // the NT every caller uses: the front end's workgroup size

// (running max, running sum of exp(x - max)) pairs of the online softmax
__device__ __forceinline__ void ms_push(float& m, float& s, float v) {
    const float mn = fmaxf(m, v);
    s = s * __expf(m - mn) + __expf(v - mn);
    m = mn;
}
__device__ __forceinline__ void ms_merge(float& m, float& s, float om, float os) {
    const float mn = fmaxf(m, om);
    s = s * __expf(m - mn) + os * __expf(om - mn);
    m = mn;
}

// One thread's share of the reduction below: thread `ft` of NT takes the groups of four logits ft, ft + NT, ..., AHEAD requests in flight, consumed in
// index order (the updates and their order do not depend on AHEAD); thread 0 also takes the up to three logits behind the last full group.
// "In flight" needs three things of the source, each learnt from the ISA (round 6: the first version had the loop, the unroll and the
// intent, and compiled to load / s_waitcnt vmcnt(0) / load / s_waitcnt ... -- ONE request in flight, 16 dependent round trips for a 128x128
// map): (1) no branch between the requests -- a request behind the end re-reads the last group and is not consumed; (2) the raw words are
// loaded and converted at the use -- a conversion next to the load is a wait next to the load; (3) the consumption is branch-free too (a
// select per request) -- with `if (...) break` the compiler sinks each load into its guarded block.
template <int NT, int AHEAD, bool VEC, typename T>
__device__ __forceinline__ void lse_share_loop(const T* __restrict__ lg, int n4, int ft, float& m, float& s) {
    for (int i = ft; i < n4; i += AHEAD * NT) {
        typename MapRaw4<T>::type r[AHEAD];
        T e[AHEAD][4];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const T* q = lg + 4 * (size_t)min(i + u * NT, n4 - 1);
            if constexpr (VEC) {
                r[u] = map_raw_load4(q);
            } else {
                e[u][0] = q[0]; e[u][1] = q[1]; e[u][2] = q[2]; e[u][3] = q[3];
            }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const bool live = i + u * NT < n4;
            float4 v;
            if constexpr (VEC) v = map_raw_cvt4<T>(r[u]);
            else v = make_float4((float)e[u][0], (float)e[u][1], (float)e[u][2], (float)e[u][3]);
            const float mn = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            const float sn = s * __expf(m - mn) + ((__expf(v.x - mn) + __expf(v.y - mn)) + (__expf(v.z - mn) + __expf(v.w - mn)));
            s = live ? sn : s;
            m = live ? mn : m;
        }
    }
}
template <int NT, int AHEAD, typename T>
__device__ __forceinline__ void lse_thread_share(const T* __restrict__ lg, int n, int ft, float& m, float& s) {
    // Groups of four consecutive logits per request, consumed in index order: the same updates in the same order as one request per
    // iteration.  A map whose sample does not start on a four-element boundary (a channel slice with H*W % 4 == 2) takes four scalar
    // loads per group instead of one vector load -- the SAME groups in the same order, so the result does not depend on where the map lies.
    const bool vec = (reinterpret_cast<uintptr_t>(lg) & (4 * sizeof(T) - 1)) == 0;
    const int n4 = n >> 2;
    if (vec) lse_share_loop<NT, AHEAD, true>(lg, n4, ft, m, s);
    else lse_share_loop<NT, AHEAD, false>(lg, n4, ft, m, s);
    if (ft == 0)  // the up to three logits behind the last full group (H*W odd: two)
        for (int i = 4 * n4; i < n; ++i) ms_push(m, s, (float)lg[i]);
}
// ... and a wavefront's: the 64 shares merged (every lane ends with the wavefront's pair)
__device__ __forceinline__ void lse_wave_merge(float& m, float& s) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) ms_merge(m, s, __shfl_xor(m, k, kWave), __shfl_xor(s, k, kWave));
}

// log-sum-exp of lg[0..n), identical in every thread of every workgroup that calls it with the same arguments; red: NT / 64 rows of LDS.
// T: the map's element type (lc_map.h) -- four elements per request whatever the type, so a 16-bit map is reduced in exactly the order
// (and to exactly the float) of the fp32 map holding the same values.
template <int NT, typename T>
__device__ __forceinline__ float block_lse(const T* __restrict__ lg, int n, float (*red)[2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -FLT_MAX, s = 0.f;
    if (tid < NT) {
        lse_thread_share<NT, LC_LSE_AHEAD>(lg, n, tid, m, s);
        lse_wave_merge(m, s);
        if (lane == 0) { red[wave][0] = m; red[wave][1] = s; }
    }
    __syncthreads();
    m = red[0][0]; s = red[0][1];
#pragma unroll
    for (int w = 1; w < NT / kWave; ++w) ms_merge(m, s, red[w][0], red[w][1]);
    return m + __logf(s);
}

}  // namespace lc
