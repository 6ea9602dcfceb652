// Stand-alone launch of the fused LC-loss forward+backward (device body: lc_loss_body.h).
#include "lc_loss_body.h"

namespace lc {
namespace {

template <bool REG, bool COV2D>
__global__ __launch_bounds__(256) void lc_cov_loss_kernel(const LossParams p) {
    __shared__ loss::LossShared sh;
    loss::sample<REG, COV2D>(p, blockIdx.x, sh);
}

}  // namespace

int launch_cov_loss(const LossParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    if (p.N <= 0) return 1;
    const int threads = p.N <= 256 ? ((p.N + 63) / 64) * 64 : 256;
    const bool reg = p.N <= 256;
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
