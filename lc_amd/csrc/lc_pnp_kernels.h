// The one-wave-per-pose kernel of the weighted-PnP solve, shared by lc_pnp.hip (large grids) and lc_pnp_latency.hip (small grids).
#pragma once
#include "lc_pnp_body.h"

namespace lc {
namespace {

// WPS = waves per SIMD the register allocator must allow: 1 for small grids (latency: B <= ~1000 poses leave most SIMDs
// idle anyway, spilling would only lengthen the lone wave), 2 for large grids (+43 % throughput at B = 16384).
// OPTS: honours PnpParams::options / weight_mask (lc_pnp_lm2_f32); the plain instantiations are the ones the metric runs
template <bool REG, int WPS, bool OPTS = false>
__global__ __launch_bounds__(64, WPS) void lc_pnp_lm_kernel(const PnpParams p) {
    __shared__ __attribute__((aligned(16))) double bc[pnp::kPnpLdsDoubles<1>];
    pnp::solve_pose<REG, 1, false, OPTS>(p, blockIdx.x, threadIdx.x, bc);
}

}  // namespace

int launch_pnp_lm_latency(const PnpParams& p, hipStream_t stream);  // N <= 64, B <= kLatencyGridMax (lc_pnp_latency.hip)
int launch_pnp_lm_chain_latency(const PnpParams& a, const PnpParams& b, int second_starts_from_first, hipStream_t stream);  // both jobs N <= 64, b.B <= kLatencyGridMax

}  // namespace lc
