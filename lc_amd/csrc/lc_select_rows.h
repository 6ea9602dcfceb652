// Row bookkeeping shared by the two kernels that compact correspondences (lc_select.hip: test-time point selection;
// lc_pnp_init.hip: the RANSAC's own inlier selection): copying one entry of a padded (B,N,.) row to its compacted slot and the
// padding rule of test.py:108-113 (fewer than `min_count` survivors out of more than `min_count` candidates: pseudo-random source
// entries, a seeded hash in the role of np.random.choice).  Device-only, no state.
#pragma once
#include "lc_common.h"

namespace lc {

__device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// test.py:108-113, called by every thread of the workgroup with the same arguments: emit(i, k) copies source entry i to slot k;
// returns the row's final count
template <class Emit>
__device__ __forceinline__ int pad_rows(int b, int n, int total, int min_count, unsigned seed, Emit&& emit) {
    if (total >= min_count || n <= min_count) return total;
    for (int k = total + (int)threadIdx.x; k < min_count; k += (int)blockDim.x)
        emit((int)(hash_u32(seed ^ hash_u32((unsigned)b * 0x9E3779B9u + (unsigned)k)) % (unsigned)n), k);
    return min_count;
}

struct RowCopy {
    const float* pts2d;   // (B,N,2)
    const float* w;       // (B,N,2)
    const float* pts3d;   // (B,N,3)
    const int* in_index;  // (B,N) or null (identity)
    float* o_pts2d;
    float* o_w;
    float* o_pts3d;
    int* o_index;         // or null
    int square;           // weights squared on the way (inv_std -> inverse covariance, test.py:92)

    // entry i of the row starting at `base` -> slot o of the same row of the outputs
    __device__ __forceinline__ void entry(size_t base, int i, int o) const {
        const float2 s = *reinterpret_cast<const float2*>(w + (base + i) * 2);
        *reinterpret_cast<float2*>(o_pts2d + (base + o) * 2) = *reinterpret_cast<const float2*>(pts2d + (base + i) * 2);
        *reinterpret_cast<float2*>(o_w + (base + o) * 2) = square ? make_float2(s.x * s.x, s.y * s.y) : s;
        const float* X = pts3d + (base + i) * 3;
        float* oX = o_pts3d + (base + o) * 3;
        oX[0] = X[0]; oX[1] = X[1]; oX[2] = X[2];
        if (o_index) o_index[base + o] = in_index ? in_index[base + i] : i;
    }

    // the same for an entry whose values the caller already holds in registers (src: its source index, in_index applied)
    __device__ __forceinline__ void entry_from(size_t base, int o, float u, float v, float2 s, float X, float Y, float Z, int src) const {
        *reinterpret_cast<float2*>(o_pts2d + (base + o) * 2) = make_float2(u, v);
        *reinterpret_cast<float2*>(o_w + (base + o) * 2) = square ? make_float2(s.x * s.x, s.y * s.y) : s;
        if (o_pts3d) {  // null: the caller fills the model points of the selected rows afterwards (lc_bits_decode_rows: code heads)
            float* oX = o_pts3d + (base + o) * 3;
            oX[0] = X; oX[1] = Y; oX[2] = Z;
        }
        if (o_index) o_index[base + o] = src;
    }

    __device__ __forceinline__ int pad(size_t base, int b, int n, int total, int min_count, unsigned seed) const {
        return pad_rows(b, n, total, min_count, seed, [&](int i, int k) { entry(base, i, k); });
    }
};

}  // namespace lc
