// On-GPU pose-error metrics: ADD, ADI, rotation error [deg], translation error for a batch of
// (estimate, ground truth) pose pairs -- lib/utils/error6d.py:87-154 (add, adi, re, te) as evaluate.py:333-339
// compute_pose_errors bundles them; replaces trimesh/cKDTree + multiprocessing.Pool(6) (evaluate.py:193-210).
// One workgroup per pose.  ADI's nearest-neighbour search is brute force over LDS tiles of the estimated-pose vertices
// (fp32 distances, fp64 means), no tree build, no host round trip.  The pair loop is the only hot part (M^2 pairs per
// pose) and is VALU-bound, so it is shaped for the packed fp32 pipe: a thread holds four gt-pose queries in registers and
// walks the tile two est-pose vertices at a time (SoA pairs in LDS, one broadcast ds_read_b64 per coordinate), so the
// three differences, the square and the two FMAs are v_pk_* instructions on (vertex j, vertex j+1) and the running
// minimum is one v_min3_f32: 3.5 VALU instructions per point pair instead of 7.
#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;   // est-pose vertices per LDS tile
constexpr int kQ = 4;         // gt-pose queries per thread
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double block_sum(double v, double* red) {
    double a[1] = {v};
    wave_allreduce<1>(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a[0];
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kThreads) void lc_pose_errors_kernel(const MetricsParams p) {
    __shared__ v2f tx[kTile / 2], ty[kTile / 2], tz[kTile / 2];  // vertex pairs (j, j+1), one array per coordinate
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
        for (int i0 = 0; i0 < M; i0 += kThreads * kQ) {  // every thread stays in the loop: the tile loads are collective
            float gx[kQ], gy[kQ], gz[kQ], best[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const int i = i0 + q * kThreads + tid;
                gx[q] = gy[q] = gz[q] = 0.f;
                best[q] = INFINITY;
                if (i < M) {
                    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                    gx[q] = Rg[0] * x + Rg[1] * y + Rg[2] * z + tg[0];
                    gy[q] = Rg[3] * x + Rg[4] * y + Rg[5] * z + tg[1];
                    gz[q] = Rg[6] * x + Rg[7] * y + Rg[8] * z + tg[2];
                }
            }
            for (int j0 = 0; j0 < M; j0 += kTile) {
                __syncthreads();
                for (int j = tid; j < kTile; j += kThreads) {
                    const int jj = j0 + j;
                    float ex = 1e30f, ey = 1e30f, ez = 1e30f;  // padding: distance^2 overflows to +inf, never the minimum
                    if (jj < M) {
                        const float x = pts[3 * jj], y = pts[3 * jj + 1], z = pts[3 * jj + 2];
                        ex = Re[0] * x + Re[1] * y + Re[2] * z + te[0];
                        ey = Re[3] * x + Re[4] * y + Re[5] * z + te[1];
                        ez = Re[6] * x + Re[7] * y + Re[8] * z + te[2];
                    }
                    reinterpret_cast<float*>(tx)[j] = ex;
                    reinterpret_cast<float*>(ty)[j] = ey;
                    reinterpret_cast<float*>(tz)[j] = ez;
                }
                __syncthreads();
                const int npair = (min(kTile, M - j0) + 1) >> 1;
#pragma unroll 4
                for (int jp = 0; jp < npair; ++jp) {
                    const v2f ex = tx[jp], ey = ty[jp], ez = tz[jp];
#pragma unroll
                    for (int q = 0; q < kQ; ++q) {
                        const v2f dx = ex - gx[q], dy = ey - gy[q], dz = ez - gz[q];
                        v2f d = dz * dz;
                        d = __builtin_elementwise_fma(dy, dy, d);
                        d = __builtin_elementwise_fma(dx, dx, d);
                        best[q] = fminf(fminf(best[q], d.x), d.y);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (i0 + q * kThreads + tid < M) adi_sum += sqrt((double)best[q]);
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
