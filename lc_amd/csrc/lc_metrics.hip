// On-GPU pose-error metrics: ADD, ADI, rotation error [deg], translation error for a batch of
// (estimate, ground truth) pose pairs -- lib/utils/error6d.py:87-154 (add, adi, re, te) as evaluate.py:333-339
// compute_pose_errors bundles them; replaces trimesh/cKDTree + multiprocessing.Pool(6) (evaluate.py:193-210).
// One workgroup per pose.  ADI's nearest-neighbour search is brute force over LDS tiles of the estimated-pose vertices
// (fp32 distances, fp64 means): M^2/256 fused-multiply-adds per thread, no tree build, no host round trip.
#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;

__device__ __forceinline__ double block_sum(double v, double* red) {
    double a[1] = {v};
    wave_allreduce<1>(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a[0];
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kThreads) void lc_pose_errors_kernel(const MetricsParams p) {
    __shared__ float4 tile[kTile];
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int off = p.pts_off ? p.pts_off[b] : 0, M = p.pts_cnt ? p.pts_cnt[b] : p.M;
    const float* pts = p.pts + 3 * (size_t)off;
    float Re[9], Rg[9], te[3], tg[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) { Re[i] = p.R_est[9 * (size_t)b + i]; Rg[i] = p.R_gt[9 * (size_t)b + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { te[i] = p.t_est[3 * (size_t)b + i]; tg[i] = p.t_gt[3 * (size_t)b + i]; }

    // ADD (error6d.py:87-101): mean || (Re p + te) - (Rg p + tg) ||
    double add_sum = 0;
    for (int i = tid; i < M; i += kThreads) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        double d2 = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double d = ((double)Re[3 * r] - Rg[3 * r]) * x + ((double)Re[3 * r + 1] - Rg[3 * r + 1]) * y +
                             ((double)Re[3 * r + 2] - Rg[3 * r + 2]) * z + ((double)te[r] - tg[r]);
            d2 += d * d;
        }
        add_sum += sqrt(d2);
    }
    add_sum = block_sum(add_sum, red);

    // ADI (error6d.py:104-125): mean over gt-pose vertices of the distance to the nearest est-pose vertex
    double adi_sum = 0;
    if (p.want_adi) {
        for (int i0 = 0; i0 < M; i0 += kThreads) {  // every thread stays in the loop: the tile loads are collective
            const int i = i0 + tid;
            const bool act = i < M;
            float gx = 0, gy = 0, gz = 0;
            if (act) {
                const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                gx = Rg[0] * x + Rg[1] * y + Rg[2] * z + tg[0];
                gy = Rg[3] * x + Rg[4] * y + Rg[5] * z + tg[1];
                gz = Rg[6] * x + Rg[7] * y + Rg[8] * z + tg[2];
            }
            float best = INFINITY;
            for (int j0 = 0; j0 < M; j0 += kTile) {
                __syncthreads();
                for (int j = tid; j < kTile; j += kThreads) {
                    const int jj = j0 + j;
                    float4 e = make_float4(1e30f, 1e30f, 1e30f, 0.f);
                    if (jj < M) {
                        const float x = pts[3 * jj], y = pts[3 * jj + 1], z = pts[3 * jj + 2];
                        e.x = Re[0] * x + Re[1] * y + Re[2] * z + te[0];
                        e.y = Re[3] * x + Re[4] * y + Re[5] * z + te[1];
                        e.z = Re[6] * x + Re[7] * y + Re[8] * z + te[2];
                    }
                    tile[j] = e;
                }
                __syncthreads();
                const int nj = min(kTile, M - j0);
#pragma unroll 8
                for (int j = 0; j < nj; ++j) {
                    const float4 e = tile[j];
                    const float dx = e.x - gx, dy = e.y - gy, dz = e.z - gz;
                    best = fminf(best, fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
                }
            }
            if (act) adi_sum += sqrt((double)best);
        }
        adi_sum = block_sum(adi_sum, red);
    }
    if (tid == 0) {
        // re (error6d.py:127-142): acos((trace(Re Rg^-1) - 1)/2) in degrees, with Rg^-1 = Rg^T for rotations -- the
        // reference inverts numerically; identical for orthonormal Rg
        double tr = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) tr += (double)Re[i] * Rg[i];
        double c = 0.5 * (tr - 1.0);
        c = fmin(1.0, fmax(-1.0, c));
        const double re_deg = acos(c) * (180.0 / 3.14159265358979323846);
        double tt = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) tt += ((double)tg[i] - te[i]) * ((double)tg[i] - te[i]);
        float* o = p.out + 4 * (size_t)b;
        o[0] = p.want_adi ? (float)(adi_sum / M) : 0.f;
        o[1] = (float)(add_sum / M);
        o[2] = (float)re_deg;
        o[3] = (float)sqrt(tt);
    }
}

}  // namespace

int launch_pose_errors(const MetricsParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_pose_errors_kernel, dim3(p.B), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc
