// "Pose unit" in ONE launch: the LC-loss forward+backward of B_loss samples and the weighted-PnP solve of B_pnp poses
// are independent (the loss linearises at the ground-truth pose, the solve starts from its own initial pose), so their
// workgroups share a grid: blocks [0, B_pnp) run pnp::solve_pose (the longer job, dispatched first), blocks
// [B_pnp, B_pnp + B_loss) run loss::sample.  At B = 256 that puts 512 single-wave workgroups on the 256 CUs at once
// instead of two half-empty launches back to back.  Only for N <= 64 (one wavefront per pose in both bodies).
#include "lc_fused_kernel.h"

namespace lc {

int launch_pose_unit(const LossParams& lp, const PnpParams& pp, hipStream_t stream) {
    if (lp.N > 64 || pp.Nmax > 64 || lp.N <= 0) return 3;
    const int blocks = (lp.B > 0 ? lp.B : 0) + (pp.B > 0 ? pp.B : 0);
    if (blocks == 0) return 0;
    if (blocks > kLatencyGridMax)
        hipLaunchKernelGGL(lc_pose_unit_kernel<LC_BIG_WPS>, dim3(blocks), dim3(64), 0, stream, lp, pp);
    else
        return launch_pose_unit_latency(lp, pp, blocks, stream);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
