// Host side of the reference ABI `pnp_ceres_f32_omp` (lib/pnp/cxx/ext.h:2-15): gather of the per-job pointer arrays into ONE
// staging buffer laid out as the zero-padded batch the kernel reads, and scatter of the results back with the reference's
// in-place rule (ceres.cpp:134-144).  Pure C++ (no HIP, no allocation): lc_capi.hip uses it on the pinned staging buffer, and
// tests/native/host_stage_sanitize.cpp runs it under AddressSanitizer + UBSan on exact-sized heap buffers (CPU only) --
// this is the one piece of host code that walks caller memory through caller-supplied counts.
#pragma once
#include <cstddef>
#include <cstring>

namespace lc {
namespace host {

constexpr int kStageSections = 8;  // K | pts3d | pts2d | sqrtL | counts | states | result_tr | rets

// byte offsets of the sections for B jobs padded to P points each (256-byte aligned); returns the total size
inline size_t stage_layout(size_t B, size_t P, size_t off[kStageSections]) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 255) & ~size_t(255); return r; };
    off[0] = take(B * 9 * 4);       // K
    off[1] = take(B * P * 3 * 4);   // pts3d
    off[2] = take(B * P * 2 * 4);   // pts2d
    off[3] = take(B * P * 4 * 4);   // sqrtL
    off[4] = take(B * 4);           // counts
    off[5] = take(B * 7 * 4);       // states
    off[6] = take(B * 4);           // result_tr
    off[7] = take(B * 4);           // rets
    return o;
}

// padded point count of a call: the largest ptCnts[i], at least 1; negative counts count as 0 (the solver flags < 3 as invalid)
inline int stage_max_points(const int* ptCnts, int B) {
    int pmax = 1;
    for (int i = 0; i < B; ++i) pmax = ptCnts[i] > pmax ? ptCnts[i] : pmax;
    return pmax;
}

// Reads exactly: 7 floats of init_states[i], 6 floats of cam_Ks[i] (ceres.cpp:99-101), ptCnts[i] x {2,3,4} floats of the point
// arrays -- and nothing of the point arrays of a job with ptCnts[i] <= 0 (their pointers may be dangling or null).
inline void stage_gather(char* h, const size_t off[kStageSections], size_t P, float* const* init_states, float* const* cam_Ks,
                         float* const* pts2ds, float* const* pts3ds, float* const* icov_sqrtLs, const int* ptCnts, int B) {
    float* hK = reinterpret_cast<float*>(h + off[0]);
    float* hX = reinterpret_cast<float*>(h + off[1]);
    float* hU = reinterpret_cast<float*>(h + off[2]);
    float* hL = reinterpret_cast<float*>(h + off[3]);
    int* hC = reinterpret_cast<int*>(h + off[4]);
    float* hS = reinterpret_cast<float*>(h + off[5]);
    for (int i = 0; i < B; ++i) {
        const size_t n = ptCnts[i] > 0 ? (size_t)ptCnts[i] : 0;
        const size_t j = (size_t)i;
        std::memcpy(hK + 9 * j, cam_Ks[i], 6 * sizeof(float));
        hK[9 * j + 6] = 0; hK[9 * j + 7] = 0; hK[9 * j + 8] = 1;
        if (n > 0) {
            std::memcpy(hX + 3 * P * j, pts3ds[i], sizeof(float) * 3 * n);
            std::memcpy(hU + 2 * P * j, pts2ds[i], sizeof(float) * 2 * n);
            std::memcpy(hL + 4 * P * j, icov_sqrtLs[i], sizeof(float) * 4 * n);
        }
        if (n < P) {
            std::memset(hX + 3 * P * j + 3 * n, 0, sizeof(float) * 3 * (P - n));
            std::memset(hU + 2 * P * j + 2 * n, 0, sizeof(float) * 2 * (P - n));
            std::memset(hL + 4 * P * j + 4 * n, 0, sizeof(float) * 4 * (P - n));
        }
        hC[i] = ptCnts[i];
        std::memcpy(hS + 7 * j, init_states[i], 7 * sizeof(float));
    }
}

// rets / result_trs for every job; init_states[i] is overwritten ONLY for converged jobs (ceres.cpp:134-144)
inline void stage_scatter(const char* h, const size_t off[kStageSections], float* const* init_states, float* result_trs, int* rets, int B) {
    const float* oS = reinterpret_cast<const float*>(h + off[5]);
    const float* oT = reinterpret_cast<const float*>(h + off[6]);
    const int* oR = reinterpret_cast<const int*>(h + off[7]);
    for (int i = 0; i < B; ++i) {
        rets[i] = oR[i];
        result_trs[i] = oT[i];
        if (oR[i] == 0) std::memcpy(init_states[i], oS + 7 * (size_t)i, 7 * sizeof(float));
    }
}

}  // namespace host
}  // namespace lc
