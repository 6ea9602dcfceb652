// "Pose unit" for the dense shapes (256 < N <= 2048) in ONE launch: the tiled LC loss (S workgroups per sample, lc_loss_tiled.h) and
// the four-wave weighted-PnP solve (one workgroup per pose, lc_pnp_body.h) are independent and both run 256-thread workgroups, so they
// share a grid -- at B = 32 the loss's 128-256 workgroups and the solve's 32 sit on different compute units at the same time instead
// of two launches back to back.  Solve workgroups first (the longer job); the loss workgroups take their (sample, slice) from the
// ticket counter as they start, so the hand-off between them stays deadlock-free whatever the dispatch order.  Same device code as the
// stand-alone launches: results bit for bit (tests/test_gpu_fused.py).
#include "lc_loss_tiled.h"
#include "lc_pnp_body.h"

namespace lc {
namespace {

union __attribute__((aligned(16))) DenseUnitShared {
    loss::LossSharedLoop loss;
    double bc[pnp::kPnpLdsDoubles<4>];
};

template <int PPT>
__global__ __launch_bounds__(256) void lc_pose_unit_dense_kernel(const LossParams lp, int T, int S, int TS, const PnpParams pp) {
    __shared__ DenseUnitShared sh;
    __shared__ unsigned ticket_sh;
    if ((int)blockIdx.x < pp.B) pnp::solve_pose<false, 4, false, false, PPT>(pp, blockIdx.x, threadIdx.x, sh.bc);
    else loss::tiled_workgroup<false>(lp, T, S, TS, sh.loss, ticket_sh, blockIdx.x - (unsigned)pp.B);
}

}  // namespace

int launch_pose_unit_dense(const LossParams& lp, const PnpParams& pp, hipStream_t stream) {
    int T, S, TS;
    if (lp.cov_2d || pp.Nmax != lp.N || pp.Nmax > 2048 || !cov_loss_tiled_shape(lp.B, lp.N, &T, &S, &TS, pp.B)) return 3;
    if (!lp.workspace || lp.workspace_bytes < cov_loss_workspace_bytes(lp.B, lp.N)) return 3;
    if (pp.options || pp.weight_mask || pp.pose_mod > 0) return 3;
    const dim3 grid((unsigned)(pp.B + lp.B * S));
    if (pp.Nmax <= 1024) hipLaunchKernelGGL(lc_pose_unit_dense_kernel<4>, grid, dim3(256), 0, stream, lp, T, S, TS, pp);
    else hipLaunchKernelGGL(lc_pose_unit_dense_kernel<8>, grid, dim3(256), 0, stream, lp, T, S, TS, pp);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
