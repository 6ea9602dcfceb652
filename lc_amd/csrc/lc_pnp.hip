// Stand-alone launch of the batched weighted-PnP solve (device body: lc_pnp_body.h).  The one-wave kernels of small grids
// (B <= kLatencyGridMax: the metric's B = 256) are instantiated in lc_pnp_latency.hip, a translation unit of its own that is compiled
// with the max-ILP machine scheduler (lc_amd/build.py: PER_FILE_FLAGS).
#include "lc_pnp_kernels.h"

namespace lc {
namespace {


// Large grids (B > kLatencyGridMax), EXPERIMENT kept for the record (profiles/r03/occupancy.txt): the low-register form of the
// same solve, three waves per SIMD (lc_pnp_body.h: solve_pose_lowreg).  Bit-identical, 168 VGPRs, no scratch -- and 18 % SLOWER
// at B = 65 536 than the two-waves-per-SIMD register build (450.6 vs 381.1 us): the reads of the LDS-parked state add ~56 waits
// on the LDS queue per iteration, which a third wave does not buy back.  Not launched unless built with -DLC_BIG_LOWREG=1.
#ifndef LC_BIG_LOWREG
#define LC_BIG_LOWREG 0
#endif
#if LC_BIG_LOWREG
template <bool OPTS = false>
__global__ __launch_bounds__(64, 3) void lc_pnp_lm_big_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLowregLdsDoubles];
    pnp::solve_pose_lowreg<OPTS>(p, blockIdx.x, threadIdx.x, bc);
}
#endif

// four wavefronts per pose for N > 64 (dense heads, ragged inference batches)
// PPT: correspondences per thread kept in registers (0: block-stride loop over memory, any N); TAIL: rows may be wider than 256 PPT
template <bool REG, bool OPTS = false, int PPT = 0, bool TAIL = false>
__global__ __launch_bounds__(256) void lc_pnp_lm_wide_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<4>];
    pnp::solve_pose<REG, 4, false, OPTS, PPT, TAIL>(p, blockIdx.x, threadIdx.x, bc);
}

// Two dependent solves in one launch (lc_pnp_lm_chain_f32): workgroup w runs pose w % a.B of the first job, then pose w of the second,
// whose start may be the first's result.  The first job's outputs are written by ONE workgroup per pose (w < a.B); a workgroup that
// repeats the solve for a further selection of the same object (w >= a.B: the solve is bit-deterministic) keeps the result to itself.
// Either way the refined state reaches the second solve through 7 floats of LDS, not through a.states in global memory -- no two
// workgroups write the same rows, nobody reads a row another workgroup is writing.
__global__ __launch_bounds__(256) void lc_pnp_lm_chain_kernel(const PnpParams a, const PnpParams b, const int second_starts_from_first) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<4>];
    __shared__ float refined[8];
    pnp::solve_pose<false, 4, false, true, 4>(a, (int)(blockIdx.x % (unsigned)a.B), threadIdx.x, bc, blockIdx.x < (unsigned)a.B, refined);
    __syncthreads();
    pnp::solve_pose<false, 4, false, true, 4>(b, blockIdx.x, threadIdx.x, bc, true, nullptr, second_starts_from_first ? refined : nullptr);
}

// Few poses, thousands of correspondences each (the test-time solves of the dense heads: 64 objects x ~3000 selected pixels): p.split_parts
// workgroups per pose, each with its share of the correspondences (lc_pnp_body.h: SPLIT, lc_common.h: block_sum_split).  The parts of a pose
// are dispatched to the same XCD (workgroups go round the 8 XCDs in launch order), so their exchange rows stay in one L2.
template <bool OPTS>
__global__ __launch_bounds__(256) void lc_pnp_lm_split_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[kSplitLdsDoubles];
    const int xcd = (int)(blockIdx.x % 8u), j = (int)(blockIdx.x / 8u);
    const int part = j % p.split_parts, b = (j / p.split_parts) * 8 + xcd;
    if (b >= p.B) return;
    SplitSum sx = split_sum_enter(static_cast<char*>(p.split_ws) + (size_t)b * kSplitPoseBytes, p.split_parts, part);
    if (threadIdx.x == 0) *reinterpret_cast<int*>(bc + 192) = 1;  // block_sum_split's "all parts arrived" (its first barrier orders this)
    pnp::solve_pose<false, 4, false, OPTS, 8, true, 1>(p, b, threadIdx.x, bc, part == 0, nullptr, nullptr, &sx);
    if (threadIdx.x == 0) split_sum_leave(sx);
}

// The launch behind every split launch (same stream: all its workgroups have ended): one workgroup per pose, which leaves at once unless a
// part of the pose gave up waiting -- then the pose's exchange region starts over from zeroes, and if the pose itself is unsolved (status
// kPnpPartNeverArrived; its row of `states` still holds the start) this workgroup solves it, playing the parts in turn: the same bits the
// split launch produces when its parts do meet (lc_common.h: block_sum_parts_serial).  Nothing flagged: ~2 us of workgroups that read two words.
template <bool OPTS>
__global__ __launch_bounds__(256) void lc_pnp_lm_split_rescue_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[kSplitSerialLdsDoubles];
    const int b = (int)blockIdx.x;
    char* region = static_cast<char*>(p.split_ws) + (size_t)b * kSplitPoseBytes;
    const bool redo = __hip_atomic_load(p.rets + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kPnpPartNeverArrived;
    if (!split_rescue_enter(region, kSplitPoseBytes, redo, (int)threadIdx.x, 256)) return;
    SplitSum sx{nullptr, nullptr, p.split_parts, 0, 0u, 0u, false};
    pnp::solve_pose<false, 4, false, OPTS, 0, false, 2>(p, b, threadIdx.x, bc, true, nullptr, nullptr, &sx);
}

// diagnostic twins that also record the per-iteration trace (tests/test_gpu_pnp_trace.py)
__global__ __launch_bounds__(64, 1) void lc_pnp_lm_trace_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<1>];
    pnp::solve_pose<true, 1, true>(p, blockIdx.x, threadIdx.x, bc);
}
__global__ __launch_bounds__(256) void lc_pnp_lm_wide_trace_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<4>];
    pnp::solve_pose<false, 4, true>(p, blockIdx.x, threadIdx.x, bc);
}

}  // namespace

int launch_pnp_lm_trace(const PnpParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.Nmax <= 64) hipLaunchKernelGGL(lc_pnp_lm_trace_kernel, dim3(p.B), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(lc_pnp_lm_wide_trace_kernel, dim3(p.B), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_pnp_lm_chain(const PnpParams& a, const PnpParams& b, hipStream_t stream) {
    const auto wide4 = [](const PnpParams& p) { return p.Nmax > 256 && p.Nmax <= 1024; };
    const float* b_start = b.start ? b.start : b.states;
    const bool rows_match = b_start != a.states || b.pose_mod == a.B || (b.pose_mod == 0 && b.B == a.B);
    // a first job in the in-place form (its states are both start and result) solved by SEVERAL second-stage workgroups would let one of
    // them read a start the other has already overwritten: one launch only when every first-stage pose has one solver, or a separate start
    const bool first_is_safe = a.start != nullptr || b.B == a.B;
    if (a.B > 0 && b.B > 0 && a.Nmax <= 64 && b.Nmax <= 64 && b.B <= kLatencyGridMax && b.B % a.B == 0 && rows_match && first_is_safe)
        return launch_pnp_lm_chain_latency(a, b, b_start == a.states ? 1 : 0, stream);  // the sparse head's two solves
    if (a.B > 0 && b.B > 0 && wide4(a) && wide4(b) && b.B % a.B == 0 && rows_match && first_is_safe) {
        hipLaunchKernelGGL(lc_pnp_lm_chain_kernel, dim3(b.B), dim3(256), 0, stream, a, b, b_start == a.states ? 1 : 0);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (int rc = launch_pnp_lm(a, stream)) return rc;
    return launch_pnp_lm(b, stream);
}

// Compute units of the current device (256 on an MI355X in SPX mode; a partitioned device reports its share), cached per device.
int device_compute_units() {
    static int cached[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && cached[dev]) return cached[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cached[dev] = n;
    return n;
}
// workgroups per unit of work for the launches whose workgroups wait for each other: the largest of 8 / 4 / 2 that keeps the grid (`units` slots,
// each workgroup a compute unit to itself) within the device -- every workgroup resident at once -- else 1
int split_parts_for(int units) {
    const int cus = device_compute_units();
    for (int parts = 8; parts >= 2; parts >>= 1)
        if ((long long)units * parts <= cus) return parts;
    return 1;
}

int pnp_split_parts(int B, int Nmax) {
    if (Nmax <= kSplitMinPoints || B <= 0) return 1;
    return split_parts_for((B + 7) / 8 * 8);  // (the grid pads the poses to a multiple of 8: lc_pnp_lm_split_kernel)
}
size_t pnp_split_workspace_bytes(int B, int Nmax) { return pnp_split_parts(B, Nmax) > 1 ? (size_t)B * kSplitPoseBytes : 0; }

int launch_pnp_lm(const PnpParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.split_ws && p.split_parts > 1) {
        const dim3 grid((unsigned)((p.B + 7) / 8 * 8 * p.split_parts));
        if (p.options || p.weight_mask || p.pose_mod > 0) {
            hipLaunchKernelGGL(lc_pnp_lm_split_kernel<true>, grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL(lc_pnp_lm_split_rescue_kernel<true>, dim3((unsigned)p.B), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL(lc_pnp_lm_split_kernel<false>, grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL(lc_pnp_lm_split_rescue_kernel<false>, dim3((unsigned)p.B), dim3(256), 0, stream, p);
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    const bool big = p.B > kLatencyGridMax;
    if (p.options || p.weight_mask || p.pose_mod > 0) {  // input filtering / weight forms folded into the load
        if (p.Nmax <= 64) {
#if LC_BIG_LOWREG
            if (big) hipLaunchKernelGGL(lc_pnp_lm_big_kernel<true>, dim3(p.B), dim3(64), 0, stream, p);
#else
            if (big) hipLaunchKernelGGL((lc_pnp_lm_kernel<true, LC_BIG_WPS, true>), dim3(p.B), dim3(64), 0, stream, p);
#endif
            else return launch_pnp_lm_latency(p, stream);
        } else if (p.Nmax <= 256) {
            hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<true, true>), dim3(p.B), dim3(256), 0, stream, p);
        } else if (p.Nmax <= 1024) {
            hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, true, 4>), dim3(p.B), dim3(256), 0, stream, p);
        } else if (p.Nmax <= 2048) {
            hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, true, 8>), dim3(p.B), dim3(256), 0, stream, p);
        } else {  // rows wider than 2048: the first 2048 correspondences of a pose in registers, the rest (if its count gets there) from memory
            hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, true, 16, true>), dim3(p.B), dim3(256), 0, stream, p);
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (p.Nmax <= 64) {
#if LC_BIG_LOWREG
        if (big) hipLaunchKernelGGL(lc_pnp_lm_big_kernel<false>, dim3(p.B), dim3(64), 0, stream, p);
#else
        if (big) hipLaunchKernelGGL((lc_pnp_lm_kernel<true, LC_BIG_WPS>), dim3(p.B), dim3(64), 0, stream, p);
#endif
        else return launch_pnp_lm_latency(p, stream);
    } else if (p.Nmax <= 256) {
        hipLaunchKernelGGL(lc_pnp_lm_wide_kernel<true>, dim3(p.B), dim3(256), 0, stream, p);
    } else if (p.Nmax <= 1024) {
        hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, false, 4>), dim3(p.B), dim3(256), 0, stream, p);
    } else if (p.Nmax <= 2048) {
        hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, false, 8>), dim3(p.B), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((lc_pnp_lm_wide_kernel<false, false, 16, true>), dim3(p.B), dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
