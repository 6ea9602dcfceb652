// The latency build of the one-wave-per-pose solve (N <= 64, grids of at most kLatencyGridMax workgroups: the metric's B = 256), in a
// translation unit of its own because it is compiled with another machine scheduler than the rest of the library:
// `-mllvm -amdgpu-sched-strategy=max-ilp` (lc_amd/build.py: PER_FILE_FLAGS).  A lone wave per SIMD has nobody to hide its latencies
// behind; the max-ILP strategy orders the ~860 instructions of an LM iteration for the shortest dependent chains instead of for
// register pressure: 13.14 -> 12.77 us for the solve, 13.52 -> 13.14 us for the pose unit, same bits (scripts/ubench/pnp_ab.py).
// At several waves per SIMD the default strategy is the faster one (B = 65 536: 671 vs 680 us), so the large-grid instantiations stay
// in lc_pnp.hip; the loss kernel alone is 4 % slower under max-ILP and stays out as well.
#include "lc_pnp_kernels.h"

namespace lc {

namespace {
// Two dependent one-wavefront solves in one launch (lc_pnp_lm_chain_f32 at the sparse head's shape: the RANSAC's inlier refinement, then the weighted
// solve on all keypoints, test.py:55-59): workgroup w solves pose w % a.B of the first job -- its outputs written by ONE workgroup per pose -- hands the
// refined state on through LDS and goes on with pose w of the second.  Same arithmetic as the two launches: same bits.
__global__ __launch_bounds__(64, 1) void lc_pnp_lm_chain_small_kernel(const PnpParams a, const PnpParams b, const int second_starts_from_first) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<1>];
    __shared__ float refined[8];
    pnp::solve_pose<true, 1, false, true>(a, (int)(blockIdx.x % (unsigned)a.B), threadIdx.x, bc, blockIdx.x < (unsigned)a.B, refined);
    __syncthreads();
    pnp::solve_pose<true, 1, false, true>(b, blockIdx.x, threadIdx.x, bc, true, nullptr, second_starts_from_first ? refined : nullptr);
}
}  // namespace

int launch_pnp_lm_chain_latency(const PnpParams& a, const PnpParams& b, int second_starts_from_first, hipStream_t stream) {
    hipLaunchKernelGGL(lc_pnp_lm_chain_small_kernel, dim3(b.B), dim3(64), 0, stream, a, b, second_starts_from_first);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_pnp_lm_latency(const PnpParams& p, hipStream_t stream) {
    if (p.options || p.weight_mask || p.pose_mod > 0) hipLaunchKernelGGL((lc_pnp_lm_kernel<true, 1, true>), dim3(p.B), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((lc_pnp_lm_kernel<true, 1>), dim3(p.B), dim3(64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
