// The metric's launch: the pose-unit kernel for grids of at most kLatencyGridMax one-wave workgroups (B = 256: 512 of them), in a
// translation unit of its own because it is compiled with the max-ILP machine scheduler (lc_amd/build.py: PER_FILE_FLAGS; measurements
// and why only here: lc_pnp_latency.hip).  The solve is the critical path of the launch; the loss half, 4 % slower under this
// scheduler when run alone, still finishes well inside it.
#include "lc_fused_kernel.h"

namespace lc {

int launch_pose_unit_latency(const LossParams& lp, const PnpParams& pp, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(lc_pose_unit_kernel<1>, dim3(blocks), dim3(64), 0, stream, lp, pp);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
