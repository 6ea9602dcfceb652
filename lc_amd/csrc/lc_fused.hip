// "Pose unit" in ONE launch: the LC-loss forward+backward of B_loss samples and the weighted-PnP solve of B_pnp poses
// are independent (the loss linearises at the ground-truth pose, the solve starts from its own initial pose), so their
// workgroups share a grid: blocks [0, B_pnp) run pnp::solve_pose (the longer job, dispatched first), blocks
// [B_pnp, B_pnp + B_loss) run loss::sample.  At B = 256 that puts 512 single-wave workgroups on the 256 CUs at once
// instead of two half-empty launches back to back.  Only for N <= 64 (one wavefront per pose in both bodies).
#include "lc_loss_body.h"
#include "lc_pnp_body.h"

#ifndef LC_UNIT_PRIO
#define LC_UNIT_PRIO 1  // A/B switch (scripts/ubench/pnp_ab.py)
#endif

#ifndef LC_UNIT_ATTR
#define LC_UNIT_ATTR
#endif

namespace lc {
namespace {

union __attribute__((aligned(16))) FusedShared {
    loss::LossShared loss;
    double bc[pnp::kPnpLdsDoubles<1>];
};

template <int WPS>  // see lc_pnp.hip: 1 = latency build for small grids, 2 = occupancy build for large ones
__global__ __launch_bounds__(64, WPS) LC_UNIT_ATTR void lc_pose_unit_kernel(const LossParams lp, const PnpParams pp) {
    __shared__ FusedShared sh;
    if ((int)blockIdx.x < pp.B) {
#if LC_UNIT_PRIO
        __builtin_amdgcn_s_setprio(3);  // the solve is the critical path of the launch: its wave wins the CU's shared issue/LDS arbitration
#endif
        pnp::solve_pose<true, 1>(pp, blockIdx.x, threadIdx.x, sh.bc);
    } else {
#ifndef LC_UNIT_NOLOSS  // diagnostic build: what does the solve cost inside this kernel without its co-runner?
        loss::sample<true>(lp, (int)blockIdx.x - pp.B, sh.loss);
#endif
    }
}

}  // namespace

int launch_pose_unit(const LossParams& lp, const PnpParams& pp, hipStream_t stream) {
    if (lp.N > 64 || pp.Nmax > 64 || lp.N <= 0) return 3;
    const int blocks = (lp.B > 0 ? lp.B : 0) + (pp.B > 0 ? pp.B : 0);
    if (blocks == 0) return 0;
    if (blocks > kLatencyGridMax)
        hipLaunchKernelGGL(lc_pose_unit_kernel<LC_BIG_WPS>, dim3(blocks), dim3(64), 0, stream, lp, pp);
    else
        hipLaunchKernelGGL(lc_pose_unit_kernel<1>, dim3(blocks), dim3(64), 0, stream, lp, pp);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
