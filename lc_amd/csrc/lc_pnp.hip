// Stand-alone launch of the batched weighted-PnP solve (device body: lc_pnp_body.h).
#include "lc_pnp_body.h"

namespace lc {
namespace {

template <bool REG>
__global__ __launch_bounds__(64) void lc_pnp_lm_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles];
    pnp::solve_pose<REG>(p, blockIdx.x, threadIdx.x, bc);
}

}  // namespace

int launch_pnp_lm(const PnpParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.Nmax <= 64)
        hipLaunchKernelGGL(lc_pnp_lm_kernel<true>, dim3(p.B), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL(lc_pnp_lm_kernel<false>, dim3(p.B), dim3(64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
