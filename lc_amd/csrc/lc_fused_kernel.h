// The pose-unit kernel (loss and solve workgroups in one grid), shared by lc_fused.hip (large grids) and lc_fused_latency.hip
// (small grids: the metric's launch, compiled with the max-ILP machine scheduler, see lc_pnp_latency.hip).
#pragma once
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

int launch_pose_unit_latency(const LossParams& lp, const PnpParams& pp, int blocks, hipStream_t stream);  // lc_fused_latency.hip

}  // namespace lc
