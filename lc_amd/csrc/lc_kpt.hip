// Keypoint negative log-likelihood of the sparse heads (`Loss_fn.sparse_kpt_loss`, losses.py:318-326):
//   nll = mean_{b,n,c} ( log sigma + |u - proj| / sigma ),  proj = project_apply(K, X, R(q), t) with the z clamp of
//   transforms.py:47-63 and the two_s = 2/|q| rotation of rotation_conversions.py:52.
// The reference runs ~25 torch ops forward (quaternion -> R, bmm, clamp, divide, abs, log, div, mean) and their autograd
// twins; the step is launch-bound (B*N*2 = 32 K elements).  Here: one launch that returns the per-sample sums and the
// unit-cotangent gradients w.r.t. pts2d and sigma, one workgroup of 64 threads per sample.
#include "lc_loss_body.h"

namespace lc {
namespace {

__global__ __launch_bounds__(64) void lc_kpt_nll_kernel(const KptParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)b * p.N;
    loss::PoseConst pc;
    {
        const float* Kp = p.K + 9 * (size_t)b;
        const float* ps = p.pose + 7 * (size_t)b;
#pragma unroll
        for (int i = 0; i < 9; ++i) pc.K[i] = Kp[i];
        const double q[4] = {ps[0], ps[1], ps[2], ps[3]};
        const double irho = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        loss::quat_matrix(q, 2.0 * irho, pc.R);
        pc.rho = 0;
        pc.t[0] = ps[4]; pc.t[1] = ps[5]; pc.t[2] = ps[6];
    }
    double acc = 0;
    for (int n = lane; n < p.N; n += kWave) {
        const float* Xp = p.pts3d + (base + n) * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        const loss::Proj pr = loss::project(pc, X);
        const float2 u = *reinterpret_cast<const float2*>(p.pts2d + (base + n) * 2);
        const float2 sg = *reinterpret_cast<const float2*>(p.std + (base + n) * 2);
        const double e0 = (double)u.x - pr.proj[0], e1 = (double)u.y - pr.proj[1];
        const double i0 = 1.0 / (double)sg.x, i1 = 1.0 / (double)sg.y;
        acc += log((double)sg.x) + fabs(e0) * i0 + log((double)sg.y) + fabs(e1) * i1;
        if (p.d_pts2d) {
            const double s0 = e0 > 0 ? 1.0 : (e0 < 0 ? -1.0 : 0.0), s1 = e1 > 0 ? 1.0 : (e1 < 0 ? -1.0 : 0.0);  // torch.sgn
            *reinterpret_cast<float2*>(p.d_pts2d + (base + n) * 2) = make_float2((float)(s0 * i0), (float)(s1 * i1));
        }
        if (p.d_std) {
            *reinterpret_cast<float2*>(p.d_std + (base + n) * 2) =
                make_float2((float)(i0 - fabs(e0) * i0 * i0), (float)(i1 - fabs(e1) * i1 * i1));
        }
    }
    double a[1] = {acc};
    wave_allreduce<1>(a);
    if (lane == 0) p.nll[b] = (float)a[0];
}

}  // namespace

int launch_kpt_nll(const KptParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_kpt_nll_kernel, dim3(p.B), dim3(kWave), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
