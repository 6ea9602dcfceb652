// Stand-alone launches of the fused LC-loss forward+backward (device body: lc_loss_body.h).
//   N <= 256                       one workgroup per sample, one correspondence per thread           lc_cov_loss_kernel<true, .>
//   N  > 256, workspace given      TILED form: one 256-thread workgroup per 4, 8 or 16 tiles of 64 correspondences, the
//                                  workgroups of a sample meet through the workspace (ONE hand-off)  lc_cov_loss_tiled_kernel
//   N  > 256 otherwise             one 256-thread workgroup per sample walking its tiles             lc_cov_loss_kernel<false, .>
// All three add the per-sample reductions in the same (tile) order: a sample's results do not depend on the form chosen.
#include "lc_loss_tiled.h"

namespace lc {
namespace {

template <bool REG, bool COV2D>
__global__ __launch_bounds__(256) void lc_cov_loss_kernel(const LossParams p) {
    using SH = typename std::conditional<REG, loss::LossShared, loss::LossSharedLoop>::type;
    __shared__ SH sh;
    loss::sample<REG, COV2D, false, SH>(p, blockIdx.x, sh);
}

// Tiled form (lc_loss_tiled.h)
template <bool COV2D>
__global__ __launch_bounds__(256) void lc_cov_loss_tiled_kernel(const LossParams p, int T, int S, int TS) {
    __shared__ loss::LossSharedLoop sh;
    __shared__ unsigned ticket_sh;
    loss::tiled_workgroup<COV2D>(p, T, S, TS, sh, ticket_sh, blockIdx.x);
}

}  // namespace

// The tiled form pays while its workgroups find a compute unit each and a sample is cut at least three ways (measured, rocprofv3,
// B x N, one-workgroup -> tiled: 32 x 1024 20.5 -> 16.1 us, 32 x 1849 37.7 -> 24.5, 64 x 2048 38.3 -> 26.8, 16 x 4096 66.8 -> 32.4,
// 64 x 4096 68.6 -> 39.9, 1 x 4096 65.9 -> 24.2; two slices only: 128 x 1024 21.6 -> 23.1; with two workgroups per unit the repeated
// quarter walk and 6x6 section of every workgroup cost more than the spread buys: 64 x 4096 at four tiles per workgroup 101 us;
// profiles/r03/tiled_loss.txt)
#ifndef LC_TILED_MAX_GROUPS
#define LC_TILED_MAX_GROUPS 256
#endif

// tiles per workgroup of the tiled form: the smallest of 4, 8, 16 that keeps the grid within LC_TILED_MAX_GROUPS; 0: loop form
static int tiled_tiles_per_group(int B, int T, int max_groups = LC_TILED_MAX_GROUPS) {
    for (int ts = 4; ts <= 16; ts *= 2) {
        const long long groups = (long long)B * ((T + ts - 1) / ts);
        if ((T + ts - 1) / ts >= 3 && groups <= max_groups) return ts;
    }
    return 0;
}

bool cov_loss_tiled_shape(int B, int N, int* T, int* S, int* TS, int reserved_groups) {
    if (B <= 0 || N <= 256) return false;
    *T = (N + loss::kTile - 1) / loss::kTile;
    *TS = tiled_tiles_per_group(B, *T, LC_TILED_MAX_GROUPS - reserved_groups);
    if (!*TS) *TS = tiled_tiles_per_group(B, *T);  // no slicing leaves that many compute units free: the stand-alone launch's
    if (!*TS) return false;
    *S = (*T + *TS - 1) / *TS;
    return true;
}

size_t cov_loss_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 256) return 0;
    const int T = (N + loss::kTile - 1) / loss::kTile;
    return tiled_tiles_per_group(B, T) ? loss::grid_workspace_bytes(B, T) : 0;
}

int launch_cov_loss(const LossParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.N <= 0) return 1;
    const bool reg = p.N <= 256;
    const size_t need = cov_loss_workspace_bytes(p.B, p.N);
    if (!reg && p.workspace && need) {
        if (p.workspace_bytes < need) return 3;
        const int T = (p.N + loss::kTile - 1) / loss::kTile, TS = tiled_tiles_per_group(p.B, T), S = (T + TS - 1) / TS;
        if (p.cov_2d) hipLaunchKernelGGL(lc_cov_loss_tiled_kernel<true>, dim3(p.B * S), dim3(256), 0, stream, p, T, S, TS);
        else hipLaunchKernelGGL(lc_cov_loss_tiled_kernel<false>, dim3(p.B * S), dim3(256), 0, stream, p, T, S, TS);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    const int threads = reg ? ((p.N + 63) / 64) * 64 : 256;
    if (p.cov_2d) {
        if (reg) hipLaunchKernelGGL((lc_cov_loss_kernel<true, true>), dim3(p.B), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((lc_cov_loss_kernel<false, true>), dim3(p.B), dim3(threads), 0, stream, p);
    } else {
        if (reg) hipLaunchKernelGGL((lc_cov_loss_kernel<true, false>), dim3(p.B), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((lc_cov_loss_kernel<false, false>), dim3(p.B), dim3(threads), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
