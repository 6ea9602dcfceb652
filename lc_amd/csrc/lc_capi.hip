// C ABI of liblc_amd.so (declared in include/lc_amd.h): argument checking, the host shim that gives the
// reference's `pnp_ceres_f32_omp` symbol a GPU body, and the tiny row-scale kernel used by autograd's chain rule.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lc_amd.h"
#include "lc_common.h"
#include <cstdint>

#include "lc_host_stage.h"

static_assert(LC_ARRIVAL_WORDS == lc::kArrivalWords, "include/lc_amd.h and lc_common.h disagree on the arrival counters");
#include "lc_kernels.h"

#ifndef LC_AMD_SRC_HASH
#define LC_AMD_SRC_HASH "unrecorded"
#endif

namespace {

// sha256 of the sources this library was compiled from, behind a marker lc_amd/build.py finds in the file's bytes (is_stale)
const char kSrcHash[] = "LC_AMD_SRC_HASH:" LC_AMD_SRC_HASH;

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define LC_HIP_OK(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(10, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

__global__ void lc_scale_rows_kernel(const float* __restrict__ scale, int B, const float* s0, float* d0, int l0,
                                     const float* s1, float* d1, int l1, const float* s2, float* d2, int l2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n0 = (size_t)B * l0, n1 = (size_t)B * l1, n2 = (size_t)B * l2;
    if (i < n0) {
        d0[i] = scale[i / l0] * s0[i];
    } else if (i < n0 + n1) {
        const size_t j = i - n0;
        d1[j] = scale[j / l1] * s1[j];
    } else if (i < n0 + n1 + n2) {
        const size_t j = i - n0 - n1;
        d2[j] = scale[j / l2] * s2[j];
    }
}

constexpr size_t kZeroCopyMaxPoints = 4096 * 64;  // measured equal-or-better up to here (scripts/ubench/host_abi_rate.py)

// Workspace of the host shim (pinned staging + device buffers), grown on demand and reused across calls.
struct PnpHostWorkspace {
    std::mutex mu;
    size_t cap_jobs = 0, cap_pts = 0;
    char* host = nullptr;
    char* host_dev = nullptr;  // device-side address of the pinned staging buffer (zero-copy route)
    char* dev = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;

    static size_t layout(size_t B, size_t P, size_t off[8]) { return lc::host::stage_layout(B, P, off); }
    int ensure(size_t B, size_t P) {
        if (!stream) LC_HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (B <= cap_jobs && P <= cap_pts) return 0;
        size_t nb = B > cap_jobs ? B : cap_jobs, np = P > cap_pts ? P : cap_pts;
        size_t off[8];
        const size_t need = layout(nb, np, off);
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        host = dev = nullptr;
        cap_jobs = cap_pts = 0;
        LC_HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&host), need, hipHostMallocMapped));
        LC_HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0));
        LC_HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev), need));
        bytes = need;
        cap_jobs = nb;
        cap_pts = np;
        return 0;
    }
};
// One workspace per DEVICE (pinned staging, device buffer and stream all belong to the device that was current when they were
// created): a process that drives several GPUs -- one thread per device, or hipSetDevice between calls -- gets the workspace of
// the device current at the call; calls on the same device serialise on its mutex, calls on different devices do not.
constexpr int kMaxDevices = 64;
PnpHostWorkspace g_ws_by_device[kMaxDevices];

int pnp_host_run(float** init_states, float** cam_Ks, float** pts2ds, float** pts3ds, float** icov_sqrtLs, int* ptCnts,
                 int maxIterCnt, float ftol, float* result_trs, int* rets, int B) {
    const int pmax = lc::host::stage_max_points(ptCnts, B);
    int device = 0;
    LC_HIP_OK(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return fail(12, "device ordinal out of range");
    PnpHostWorkspace& g_ws = g_ws_by_device[device];
    std::lock_guard<std::mutex> lock(g_ws.mu);
    if (int rc = g_ws.ensure(B, pmax)) return rc;
    const size_t P = pmax;
    size_t off[8];
    const size_t used = PnpHostWorkspace::layout(B, P, off);
    char* h = g_ws.host;
    lc::host::stage_gather(h, off, P, init_states, cam_Ks, pts2ds, pts3ds, icov_sqrtLs, ptCnts, B);  // lc_host_stage.h (sanitised on the CPU)
    // Small batches: the kernel reads the pinned staging buffer and writes its results there directly over PCIe (no copy
    // commands: two API calls and their DMA set-up cost more than the ~2 KB/pose they move); large ones: one H2D, one D2H.
    static const int zc_env = [] { const char* e = std::getenv("LC_AMD_HOST_ZEROCOPY"); return e ? std::atoi(e) : -1; }();
    const bool zero_copy = zc_env >= 0 ? zc_env != 0 : (size_t)B * P <= kZeroCopyMaxPoints;
    char* d = zero_copy ? g_ws.host_dev : g_ws.dev;
    hipStream_t st = g_ws.stream;
    if (!zero_copy) LC_HIP_OK(hipMemcpyAsync(d, h, off[6], hipMemcpyHostToDevice, st));  // everything up to and including states
    lc::PnpParams p{};
    p.K = reinterpret_cast<float*>(d + off[0]);
    p.pts3d = reinterpret_cast<float*>(d + off[1]);
    p.pts2d = reinterpret_cast<float*>(d + off[2]);
    p.sqrtL = reinterpret_cast<float*>(d + off[3]);
    p.sqrt_diag = nullptr;
    p.counts = reinterpret_cast<int*>(d + off[4]);
    p.start = nullptr;
    p.states = reinterpret_cast<float*>(d + off[5]);
    p.result_tr = reinterpret_cast<float*>(d + off[6]);
    p.rets = reinterpret_cast<int*>(d + off[7]);
    p.iters = nullptr;
    p.B = B; p.Nmax = (int)P; p.max_iter = maxIterCnt; p.ftol = ftol;
    if (lc::launch_pnp_lm(p, st)) return fail(11, "pnp kernel launch failed");
    if (!zero_copy) LC_HIP_OK(hipMemcpyAsync(h + off[5], d + off[5], used - off[5], hipMemcpyDeviceToHost, st));
    LC_HIP_OK(hipStreamSynchronize(st));
    lc::host::stage_scatter(h, off, init_states, result_trs, rets, B);
    return 0;
}

}  // namespace

// Pointers the kernels access with 8-byte (float2 rows: pts2d, inv_std, their gradients) or 16-byte (float4: 2x2 factors, head
// and code maps) vector instructions.  Row slices of contiguous batches always satisfy this; an arbitrary element offset may not.
template <typename... P>
static bool misaligned(size_t bytes, P... ptrs) {
    const void* v[] = {static_cast<const void*>(ptrs)...};
    for (const void* q : v)
        if (q && (reinterpret_cast<uintptr_t>(q) & (bytes - 1))) return true;
    return false;
}
#define LC_REQUIRE_ALIGNED(bytes, ...) \
    if (misaligned(bytes, __VA_ARGS__)) return fail(1, "pointer not " #bytes "-byte aligned (" #__VA_ARGS__ ")")

#pragma GCC visibility push(default)
extern "C" {

int lc_amd_version(void) { return LC_AMD_VERSION; }
const char* lc_amd_source_hash(void) { return kSrcHash + sizeof("LC_AMD_SRC_HASH:") - 1; }
const char* lc_amd_last_error(void) { return g_err.c_str(); }

void pnp_ceres_f32_omp(float** init_states, float** cam_Ks, float** pts2ds, float** pts3ds, float** icov_sqrtLs,
                       int* ptCnts, int maxIterCnt, float function_tolerance, int printSummary, float* result_trs,
                       int* rets, int job_count, int num_threads) {
    (void)num_threads;
    if (job_count <= 0) return;
    const int rc = pnp_host_run(init_states, cam_Ks, pts2ds, pts3ds, icov_sqrtLs, ptCnts, maxIterCnt, function_tolerance,
                                result_trs, rets, job_count);
    if (rc != 0) {
        std::fprintf(stderr, "lc_amd: pnp_ceres_f32_omp failed on the GPU path (%s); marking %d jobs invalid\n",
                     g_err.c_str(), job_count);
        for (int i = 0; i < job_count; ++i) { rets[i] = 1; result_trs[i] = 1.f; }
        return;
    }
    if (printSummary) {
        for (int i = 0; i < job_count; ++i)
            std::printf("lc_amd pnp job %d: points=%d invalid=%d trust_region_radius=%g\n", i, ptCnts[i], rets[i], result_trs[i]);
    }
}

// argument checks of lc_pnp_lm3_f32 -> kernel parameters; 0 / 1 (lc_amd_last_error says why)
static int pnp_params(const float* K, const float* pts3d, const float* pts2d, const float* sqrtL, const float* weights_diag,
                      const unsigned char* weight_mask, const int* counts, const float* start, float* states, float* result_tr, int* rets,
                      int* iters, int B, int Nmax, int max_iter, float function_tolerance, int options, int pose_mod, lc::PnpParams& p) {
    if (B < 0 || Nmax < 0 || pose_mod < 0) return fail(1, "negative size");
    if (B == 0) {  // an empty batch is a no-op whatever else is passed
        p = lc::PnpParams{};
        return 0;
    }
    if ((sqrtL != nullptr) + (weights_diag != nullptr) + (weight_mask != nullptr) != 1)
        return fail(1, "exactly one of sqrtL / weights_diag / weight_mask must be given");
    if (options & ~(LC_PNP_WEIGHTS_ARE_ICOV | LC_PNP_NAN_TO_NUM | LC_PNP_WEIGHTS_ARE_STD)) return fail(1, "unknown option bit");
    if ((options & LC_PNP_WEIGHTS_ARE_ICOV) && !weights_diag) return fail(1, "LC_PNP_WEIGHTS_ARE_ICOV needs weights_diag");
    if ((options & LC_PNP_WEIGHTS_ARE_STD) && !(options & LC_PNP_WEIGHTS_ARE_ICOV)) return fail(1, "LC_PNP_WEIGHTS_ARE_STD goes with LC_PNP_WEIGHTS_ARE_ICOV (deviations -> inverse variances -> factor)");
    if (pose_mod > 0 && (!start || start == states)) return fail(1, "pose_mod needs a separate start array");
    if (!K || !pts3d || !pts2d || !states || !result_tr || !rets) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, pts2d, weights_diag);
    LC_REQUIRE_ALIGNED(16, sqrtL);
    p = lc::PnpParams{K, pts2d, pts3d, sqrtL, weights_diag, counts, start == states ? nullptr : start, states, result_tr, rets, iters,
                      B, Nmax, max_iter, function_tolerance, nullptr, 0, options, weight_mask, pose_mod};
    return 0;
}

// hands a job the caller's workspace when its shape takes the split form (lc_kernels.h: pnp_split_parts); 0 / 1
static int pnp_attach_workspace(lc::PnpParams& p, void* workspace, size_t workspace_bytes) {
    if (!workspace || p.B <= 0) return 0;
    const size_t need = lc::pnp_split_workspace_bytes(p.B, p.Nmax);
    if (need == 0) return 0;
    if (workspace_bytes < need) return fail(1, "pnp workspace smaller than lc_pnp_lm_workspace_bytes(B, Nmax)");
    if (reinterpret_cast<uintptr_t>(workspace) & 127u) return fail(1, "pnp workspace must be 128-byte aligned");
    p.split_ws = workspace;
    p.split_parts = lc::pnp_split_parts(p.B, p.Nmax);
    return 0;
}

size_t lc_pnp_lm_workspace_bytes(int B, int Nmax) { return lc::pnp_split_workspace_bytes(B, Nmax); }

int lc_pnp_lm3_f32(const float* K, const float* pts3d, const float* pts2d, const float* sqrtL, const float* weights_diag,
                   const unsigned char* weight_mask, const int* counts, const float* start, float* states, float* result_tr, int* rets,
                   int* iters, int B, int Nmax, int max_iter, float function_tolerance, int options, int pose_mod, void* workspace,
                   size_t workspace_bytes, void* stream) {
    lc::PnpParams p;
    if (int rc = pnp_params(K, pts3d, pts2d, sqrtL, weights_diag, weight_mask, counts, start, states, result_tr, rets, iters, B, Nmax, max_iter,
                            function_tolerance, options, pose_mod, p))
        return rc;
    if (B == 0) return 0;
    if (int rc = pnp_attach_workspace(p, workspace, workspace_bytes)) return rc;
    if (lc::launch_pnp_lm(p, static_cast<hipStream_t>(stream))) return fail(11, "pnp kernel launch failed");
    return 0;
}

long long lc_split_workspace_rescues(const void* workspace, size_t workspace_bytes, int kind, void* stream) {
    // per-unit region size of the kind: what one unit of a shape that takes the split form asks for
    const size_t unit = kind == 0 ? lc::pnp_split_workspace_bytes(1, 4 * lc::kSplitMinPoints) : kind == 1 ? lc::dense_select_split_workspace_bytes(1, 16384) : 0;
    if (!workspace || unit < 128 || workspace_bytes < unit) { fail(1, "lc_split_workspace_rescues: kind 0 (pnp) | 1 (select) and a workspace of at least one unit"); return -1; }
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) { fail(11, "stream synchronisation failed"); return -1; }
    long long total = 0;
    for (size_t u = 0; (u + 1) * unit <= workspace_bytes; ++u) {
        unsigned tail[3];  // epoch, dirty, rescues so far (lc_common.h: the last 128 bytes of a unit's region)
        if (hipMemcpy(tail, static_cast<const char*>(workspace) + (u + 1) * unit - 128, sizeof(tail), hipMemcpyDeviceToHost) != hipSuccess) { fail(11, "copy failed"); return -1; }
        total += tail[2];
    }
    return total;
}

int lc_pnp_lm_chain2_f32(const lc_pnp_lm_job* first, const lc_pnp_lm_job* second, void* workspace, size_t workspace_bytes, void* stream) {
    if (!first || !second) return fail(1, "null job");
    lc::PnpParams a, b;
    const lc_pnp_lm_job* jobs[2] = {first, second};
    lc::PnpParams* ps[2] = {&a, &b};
    for (int k = 0; k < 2; ++k) {
        const lc_pnp_lm_job& j = *jobs[k];
        if (int rc = pnp_params(j.K, j.pts3d, j.pts2d, j.sqrtL, j.weights_diag, j.weight_mask, j.counts, j.start, j.states, j.result_tr, j.rets,
                                j.iters, j.B, j.Nmax, j.max_iter, j.function_tolerance, j.options, j.pose_mod, *ps[k]))
            return rc;
        if (int rc = pnp_attach_workspace(*ps[k], workspace, workspace_bytes)) return rc;  // the two launches are ordered: one workspace serves both
    }
    if (lc::launch_pnp_lm_chain(a, b, static_cast<hipStream_t>(stream))) return fail(11, "pnp kernel launch failed");
    return 0;
}

int lc_pnp_lm_trace_f32(const float* K, const float* pts3d, const float* pts2d, const float* sqrtL, const float* sqrt_diag,
                        const int* counts, const float* start, float* states, float* result_tr, int* rets, int* iters, int B, int Nmax,
                        int max_iter, float function_tolerance, double* trace, int trace_rows, void* stream) {
    if (B < 0 || Nmax < 0 || trace_rows < 0) return fail(1, "negative size");
    if (B == 0) return 0;
    if ((sqrtL == nullptr) == (sqrt_diag == nullptr)) return fail(1, "exactly one of sqrtL / sqrt_diag must be given");
    if (!K || !pts3d || !pts2d || !states || !result_tr || !rets || !trace) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, pts2d, sqrt_diag, trace);
    LC_REQUIRE_ALIGNED(16, sqrtL);
    lc::PnpParams p{K, pts2d, pts3d, sqrtL, sqrt_diag, counts, start == states ? nullptr : start, states, result_tr, rets, iters,
                    B, Nmax, max_iter, function_tolerance, trace, trace_rows};
    if (lc::launch_pnp_lm_trace(p, static_cast<hipStream_t>(stream))) return fail(11, "pnp trace kernel launch failed");
    return 0;
}

size_t lc_cov_loss_workspace_bytes(int B, int N) { return lc::cov_loss_workspace_bytes(B, N); }

int lc_cov_loss3_fwd_bwd_f32(const float* K, const float* pose, const float* pts3d, const float* pts2d, const float* inv_std,
                             const float* valid, const float* bbox_3d, const float* grad_out, int B, int N, float max_err_len,
                             float rel_thresh, float w_e_thresh, int cov_2d, float* loss, float* d_pts2d, float* d_inv_std,
                             float* d_pts3d, float* aux, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || N <= 0) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!K || !pose || !pts3d || !pts2d || !inv_std || !bbox_3d || !loss) return fail(1, "null pointer");
    if ((d_pts2d == nullptr) != (d_inv_std == nullptr)) return fail(1, "d_pts2d and d_inv_std must both be given or both be NULL");
    if (d_pts3d && !d_pts2d) return fail(1, "d_pts3d needs d_pts2d/d_inv_std");
    LC_REQUIRE_ALIGNED(8, pts2d, inv_std, d_pts2d, d_inv_std);
    LC_REQUIRE_ALIGNED(256, workspace);
    lc::LossParams p{K, pose, pts3d, pts2d, inv_std, valid, bbox_3d, grad_out, loss, d_pts2d, d_inv_std, d_pts3d, aux,
                     B, N, max_err_len, rel_thresh, w_e_thresh, cov_2d ? 1 : 0, workspace, workspace ? workspace_bytes : 0};
    const int rc = lc::launch_cov_loss(p, static_cast<hipStream_t>(stream));
    if (rc == 3) return fail(3, "workspace smaller than lc_cov_loss_workspace_bytes(B, N)");
    if (rc) return fail(11, "loss kernel launch failed");
    return 0;
}

int lc_pose_unit2_f32(const float* K, const float* pose, const float* pts3d, const float* pts2d, const float* inv_std,
                      const float* valid, const float* bbox_3d, const float* grad_out, int B, int N, float max_err_len,
                      float rel_thresh, float w_e_thresh, float* loss, float* d_pts2d, float* d_inv_std, float* d_pts3d,
                      const float* pnp_sqrt_diag, const float* pnp_start, float* pnp_states, float* pnp_result_tr, int* pnp_rets,
                      int* pnp_iters, int pnp_max_iter, float pnp_function_tolerance, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || N <= 0) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!K || !pose || !pts3d || !pts2d || !inv_std || !bbox_3d || !loss || !d_pts2d || !d_inv_std || !pnp_sqrt_diag ||
        !pnp_start || !pnp_states || !pnp_result_tr || !pnp_rets)
        return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, pts2d, inv_std, d_pts2d, d_inv_std, pnp_sqrt_diag);
    LC_REQUIRE_ALIGNED(256, workspace);  // one rule for the tiled loss's workspace, here and in lc_cov_loss3_fwd_bwd_f32
    lc::LossParams lp{K, pose, pts3d, pts2d, inv_std, valid, bbox_3d, grad_out, loss, d_pts2d, d_inv_std, d_pts3d, nullptr,
                      B, N, max_err_len, rel_thresh, w_e_thresh, 0, workspace, workspace ? workspace_bytes : 0};
    lc::PnpParams pp{K, pts2d, pts3d, nullptr, pnp_sqrt_diag, nullptr, pnp_start == pnp_states ? nullptr : pnp_start, pnp_states,
                     pnp_result_tr, pnp_rets, pnp_iters, B, N, pnp_max_iter, pnp_function_tolerance};
    const int rc = N <= 64 ? lc::launch_pose_unit(lp, pp, static_cast<hipStream_t>(stream)) : lc::launch_pose_unit_dense(lp, pp, static_cast<hipStream_t>(stream));
    if (rc == 3) return fail(3, "lc_pose_unit2_f32 takes N <= 64, or 256 < N <= 2048 with the workspace of lc_cov_loss_workspace_bytes(B, N) "
                                "where that is non-zero; launch the two kernels separately otherwise");
    if (rc) return fail(11, "pose-unit kernel launch failed");
    return 0;
}

int lc_scale_rows_f32(const float* scale, int B, const float* src0, float* dst0, int len0, const float* src1, float* dst1,
                      int len1, const float* src2, float* dst2, int len2, void* stream) {
    if (B <= 0) return 0;
    if (!src0) len0 = 0;
    if (!src1) len1 = 0;
    if (!src2) len2 = 0;
    const size_t total = (size_t)B * ((size_t)len0 + len1 + len2);
    if (total == 0) return 0;
    const int threads = 256;
    const size_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(lc_scale_rows_kernel, dim3((unsigned)blocks), dim3(threads), 0, static_cast<hipStream_t>(stream), scale, B,
                       src0, dst0, len0, src1, dst1, len1, src2, dst2, len2);  // a zero length makes its branch unreachable
    return hipGetLastError() == hipSuccess ? 0 : fail(11, "scale kernel launch failed");
}

static int head_fwd(const void* in, int dtype, int M, int H, int W, int is_prob, float* mean, float* std, float* stats, void* stream) {
    if (M < 0 || H <= 0 || W <= 0) return fail(1, "bad size");
    if (dtype < 0 || dtype > 2) return fail(1, "dtype must be LC_F32, LC_F16 or LC_BF16");
    if (M == 0) return 0;
    if (!in || !mean || !std || !stats) return fail(1, "null pointer");
    if (misaligned(dtype == 0 ? 4 : 2, in)) return fail(1, "map pointer not aligned to its element type");
    lc::HeadParams p{in, mean, std, stats, M, H, W, is_prob, dtype};
    const int rc = lc::launch_head_fwd(p, static_cast<hipStream_t>(stream));
    if (rc == 3) return fail(3, "map too large for the single-pass soft-argmax kernel");
    return rc ? fail(11, "head kernel launch failed") : 0;
}

static int head_bwd(const void* in, int dtype, const float* mean, const float* std, const float* stats, const float* g_mean,
                    const float* g_std, int M, int H, int W, int is_prob, void* g_in, void* stream) {
    if (M < 0 || H <= 0 || W <= 0) return fail(1, "bad size");
    if (dtype < 0 || dtype > 2) return fail(1, "dtype must be LC_F32, LC_F16 or LC_BF16");
    if (M == 0) return 0;
    if (!in || !mean || !std || !stats || !g_mean || !g_std || !g_in) return fail(1, "null pointer");
    if (misaligned(dtype == 0 ? 4 : 2, in, g_in)) return fail(1, "map pointer not aligned to its element type");
    lc::HeadBwdParams p{in, mean, std, stats, g_mean, g_std, g_in, M, H, W, is_prob, dtype};
    return lc::launch_head_bwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "head backward launch failed") : 0;
}

int lc_softargmax2d_fwd(const void* in, int dtype, int M, int H, int W, int is_prob, float* mean, float* std, float* stats, void* stream) {
    return head_fwd(in, dtype, M, H, W, is_prob, mean, std, stats, stream);
}

int lc_softargmax2d_bwd(const void* in, int dtype, const float* mean, const float* std, const float* stats, const float* g_mean,
                        const float* g_std, int M, int H, int W, int is_prob, void* g_in, void* stream) {
    return head_bwd(in, dtype, mean, std, stats, g_mean, g_std, M, H, W, is_prob, g_in, stream);
}

int lc_dense_frontend_fwd3(const void* xyz, const void* wlogits, const void* wscale, const float* noc_scale, const void* vis_logits,
                               float vis_thresh, int map_dtype, int xyz_dtype, int wscale_dtype, long long xyz_bstride, long long wlogits_bstride, long long vis_bstride, int B, int H, int W, int top, int left, int sample, float* pts2d, float* inv_std,
                               float* pts3d, float* lse, unsigned char* vis_mask, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (wscale_dtype < 0 || wscale_dtype > 2) return fail(1, "wscale_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (xyz_dtype != map_dtype && xyz_dtype != 0) return fail(1, "xyz_dtype must be map_dtype or LC_F32");
    if (xyz_bstride < 0 || wlogits_bstride < 0 || vis_bstride < 0) return fail(1, "negative batch stride");
    if ((xyz_bstride && xyz_bstride < 3ll * H * W) || (wlogits_bstride && wlogits_bstride < 2ll * H * W) || (vis_bstride && vis_bstride < 1ll * H * W)) return fail(1, "batch stride smaller than a sample");
    if (B < 0 || H <= 0 || W <= 0 || sample <= 0 || top < 0 || left < 0 || top >= H || left >= W) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!wlogits || !wscale || !pts2d || !inv_std || !lse || (xyz != nullptr) != (pts3d != nullptr)) return fail(1, "null pointer");
    if ((vis_logits != nullptr) != (vis_mask != nullptr)) return fail(1, "vis_logits and vis_mask go together");
    LC_REQUIRE_ALIGNED(8, pts2d, inv_std);
    const int N = ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample);
    lc::DenseParams p{xyz, wlogits, wscale, noc_scale, pts2d, inv_std, pts3d, lse, B, H, W, N, top, left, sample, vis_logits, vis_thresh, vis_mask, map_dtype, wscale_dtype, xyz_dtype,
                       xyz_bstride ? xyz_bstride : 3ll * H * W, wlogits_bstride ? wlogits_bstride : 2ll * H * W, vis_bstride ? vis_bstride : 1ll * H * W};
    return lc::launch_dense_fwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "dense front-end launch failed") : 0;
}

int lc_dense_frontend_bwd2(const void* wlogits, const void* wscale, const float* noc_scale, const float* lse,
                              const float* g_inv_std, const float* g_pts3d, int map_dtype, int wscale_dtype, long long wlogits_bstride, int B, int H, int W, int top, int left, int sample,
                              void* d_xyz, void* d_wlogits, void* d_wscale, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (wscale_dtype < 0 || wscale_dtype > 2) return fail(1, "wscale_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (wlogits_bstride < 0) return fail(1, "negative batch stride");
    if (wlogits_bstride && wlogits_bstride < 2ll * H * W) return fail(1, "batch stride smaller than a sample");
    if (B < 0 || H <= 0 || W <= 0 || sample <= 0 || top < 0 || left < 0 || top >= H || left >= W) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!wlogits || !wscale || !lse) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, g_inv_std);
    const int N = ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample);
    lc::DenseBwdParams p{wlogits, wscale, noc_scale, lse, g_inv_std, g_pts3d, d_xyz, d_wlogits, d_wscale, B, H, W, N, top, left, sample, map_dtype, wscale_dtype, wlogits_bstride ? wlogits_bstride : 2ll * H * W};
    return lc::launch_dense_bwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "dense front-end backward launch failed") : 0;
}

size_t lc_pnp_ransac_workspace_bytes(int B, int Nmax, int iterations) {
    if (B <= 0 || Nmax < 0 || iterations <= 0) return 0;
    return lc::pnp_ransac_workspace_bytes(B, Nmax, (iterations + 63) / 64);
}

int lc_pnp_ransac_workspace_layout(int B, int Nmax, int iterations, size_t out[6]) {
    if (B <= 0 || Nmax < 0 || iterations <= 0 || !out) return fail(1, "bad size");
    lc::pnp_ransac_workspace_layout(B, Nmax, (iterations + 63) / 64, out);
    return 0;
}

static int ransac_init_any(const float* K, const float* pts3d, const float* pts2d, const int* counts, int B, int Nmax,
                           float reproj_err, const float* reproj_err_per_pose, int iterations, unsigned seed, float* states,
                           unsigned char* inlier_mask, int* n_inliers, int* invalid, int* best_hyp, int* valid_counts, void* workspace,
                           size_t workspace_bytes, int ticketed, const float* sel_w, const int* sel_in_index, int sel_min_count, unsigned sel_seed,
                           float* sel_pts2d, float* sel_w_out, float* sel_pts3d, int* sel_index, int* sel_counts, int pose_index_offset, int per_pose_divides,
                           void* stream) {
    if (B < 0 || Nmax < 0 || iterations <= 0 || sel_min_count < 0) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!K || !pts3d || !pts2d || !states || !inlier_mask || !n_inliers || !invalid) return fail(1, "null pointer");
    if (sel_w && (!sel_pts2d || !sel_w_out || !sel_pts3d || !sel_counts)) return fail(1, "null selection output");
    LC_REQUIRE_ALIGNED(16, workspace);  // the selection reads the hypotheses as 16-byte pairs of doubles
    LC_REQUIRE_ALIGNED(8, sel_w);
    LC_REQUIRE_ALIGNED(8, sel_w_out);
    LC_REQUIRE_ALIGNED(8, sel_pts2d);
    LC_REQUIRE_ALIGNED(8, pts2d);
    lc::RansacParams p{K, pts3d, pts2d, counts, reproj_err_per_pose, states, inlier_mask, n_inliers, invalid, B, Nmax,
                       (iterations + 63) / 64, reproj_err, seed, best_hyp, valid_counts, workspace, workspace_bytes,
                       sel_w, sel_in_index, sel_pts2d, sel_w_out, sel_pts3d, sel_index, sel_counts, sel_min_count, sel_seed, (workspace && ticketed) ? 1 : 0, pose_index_offset,
                       per_pose_divides};
    const int rc = lc::launch_pnp_ransac(p, static_cast<hipStream_t>(stream));
    if (rc == 3) return fail(1, "workspace smaller than lc_pnp_ransac_workspace_bytes(B, Nmax, iterations)");
    return rc ? fail(11, "ransac kernel launch failed") : 0;
}

int lc_pnp_ransac_init5_f32(const float* K, const float* pts3d, const float* pts2d, const int* counts, int B, int Nmax,
                            float reproj_err, const float* reproj_err_per_pose, int iterations, unsigned seed, float* states,
                            unsigned char* inlier_mask, int* n_inliers, int* invalid, int* best_hyp, int* valid_counts, void* workspace,
                            size_t workspace_bytes, int ticketed, const float* sel_w, const int* sel_in_index, int sel_min_count, unsigned sel_seed,
                            float* sel_pts2d, float* sel_w_out, float* sel_pts3d, int* sel_index, int* sel_counts, int pose_index_offset, void* stream) {
    // a positive scalar next to per-pose values means "divide" (include/lc_amd.h); a scalar <= 0: the per-pose values are the thresholds
    return ransac_init_any(K, pts3d, pts2d, counts, B, Nmax, reproj_err, reproj_err_per_pose, iterations, seed, states, inlier_mask, n_inliers, invalid, best_hyp,
                           valid_counts, workspace, workspace_bytes, ticketed, sel_w, sel_in_index, sel_min_count, sel_seed, sel_pts2d, sel_w_out, sel_pts3d, sel_index,
                           sel_counts, pose_index_offset, (reproj_err_per_pose && reproj_err > 0.f) ? 1 : 0, stream);
}

static int bits_check(int B, int C, int H, int W, int n0, int n1, int n2, int top, int left, int sample) {
    if (B < 0 || H <= 0 || W <= 0 || sample <= 0 || top < 0 || left < 0 || top >= H || left >= W) return fail(1, "bad size");
    if (n0 < 1 || n1 < 1 || n2 < 1 || n0 > 24 || n1 > 24 || n2 > 24 || n0 + n1 + n2 != C) return fail(1, "bad bit counts");
    return 0;
}

int lc_bits_decode_gt_fwd3(const void* logits, const unsigned char* gt_bits, const unsigned char* gt_msk, const float* out_scale,
                               const float* out_xform, int map_dtype, long long logits_bstride, int B, int C, int H, int W, int n0, int n1, int n2, int black_background, int top,
                               int left, int sample, float* out, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0) return fail(1, "negative batch stride");
    if (logits_bstride && logits_bstride < (long long)C * H * W) return fail(1, "batch stride smaller than a sample");
    if (int rc = bits_check(B, C, H, W, n0, n1, n2, top, left, sample)) return rc;
    if (B == 0) return 0;
    if (!logits || !gt_bits || !out) return fail(1, "null pointer");
    if (out_xform && !out_scale) return fail(1, "the model transform applies to scaled coordinates: out_scale is needed with out_xform");
    const int N = ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample);
    lc::BitsParams p{logits, gt_bits, gt_msk, nullptr, out, nullptr, B, C, H, W, N, top, left, sample, {n0, n1, n2},
                     black_background ? -1 : 1, out_scale, out_xform, 0, map_dtype, logits_bstride ? logits_bstride : (long long)C * H * W};
    return lc::launch_bits_decode_gt_fwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "bits decode launch failed") : 0;
}

int lc_bits_decode_gt_bwd3(const void* logits, const unsigned char* gt_bits, const unsigned char* gt_msk, const float* out_scale,
                               const float* out_xform, const float* g_out, int map_dtype, long long logits_bstride, int B, int C, int H, int W, int n0, int n1, int n2,
                               int black_background, int top, int left, int sample, void* d_logits, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0) return fail(1, "negative batch stride");
    if (logits_bstride && logits_bstride < (long long)C * H * W) return fail(1, "batch stride smaller than a sample");
    if (int rc = bits_check(B, C, H, W, n0, n1, n2, top, left, sample)) return rc;
    if (B == 0) return 0;
    if (!logits || !gt_bits || !g_out || !d_logits) return fail(1, "null pointer");
    if (out_xform && !out_scale) return fail(1, "the model transform applies to scaled coordinates: out_scale is needed with out_xform");
    const int N = ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample);
    lc::BitsParams p{logits, gt_bits, gt_msk, g_out, nullptr, d_logits, B, C, H, W, N, top, left, sample, {n0, n1, n2},
                     black_background ? -1 : 1, out_scale, out_xform, 0, map_dtype, logits_bstride ? logits_bstride : (long long)C * H * W};
    return lc::launch_bits_decode_gt_bwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "bits decode backward launch failed") : 0;
}

int lc_bits_decode3(const void* logits, const float* out_scale, const float* out_xform, int map_dtype, long long logits_bstride, int B, int C, int H, int W, int n0, int n1, int n2,
                        int black_background, int planar, float* out, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0) return fail(1, "negative batch stride");
    if (logits_bstride && logits_bstride < (long long)C * H * W) return fail(1, "batch stride smaller than a sample");
    if (int rc = bits_check(B, C, H, W, n0, n1, n2, 0, 0, 1)) return rc;
    if (B == 0) return 0;
    if (!logits || !out) return fail(1, "null pointer");
    if (out_xform && !out_scale) return fail(1, "the model transform applies to scaled coordinates: out_scale is needed with out_xform");
    lc::BitsParams p{logits, nullptr, nullptr, nullptr, out, nullptr, B, C, H, W, H * W, 0, 0, 1, {n0, n1, n2}, black_background ? -1 : 1,
                     out_scale, out_xform, planar ? 1 : 0, map_dtype, logits_bstride ? logits_bstride : (long long)C * H * W};
    return lc::launch_bits_decode(p, static_cast<hipStream_t>(stream)) ? fail(11, "bits decode launch failed") : 0;
}

int lc_bits_decode_rows(const void* logits, const float* out_scale, const float* out_xform, int map_dtype, long long logits_bstride, int B, int C, int H,
                        int W, int n0, int n1, int n2, int black_background, int top, int left, int sample, const int* rows_index,
                        const int* rows_counts, int rows_N, float* out_pts3d, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0 || (logits_bstride && logits_bstride < (long long)C * H * W)) return fail(1, "batch stride smaller than a sample");
    if (int rc = bits_check(B, C, H, W, n0, n1, n2, top, left, sample)) return rc;
    if (rows_N < 0) return fail(1, "bad size");
    if (B == 0 || rows_N == 0) return 0;
    if (B > 65535) return fail(1, "more than 65535 rows");
    if (!logits || !rows_index || !rows_counts || !out_pts3d) return fail(1, "null pointer");
    if (out_xform && !out_scale) return fail(1, "the model transform applies to scaled coordinates: out_scale is needed with out_xform");
    lc::BitsParams p{logits, nullptr, nullptr, nullptr, out_pts3d, nullptr, B, C, H, W, H * W, top, left, sample, {n0, n1, n2}, black_background ? -1 : 1,
                     out_scale, out_xform, 0, map_dtype, logits_bstride ? logits_bstride : (long long)C * H * W, rows_index, rows_counts, rows_N};
    return lc::launch_bits_decode_rows(p, static_cast<hipStream_t>(stream)) ? fail(11, "bits decode (rows) launch failed") : 0;
}

int lc_pose_errors_f32(const float* R_est, const float* t_est, const float* R_gt, const float* t_gt, const float* pts,
                       const int* pts_off, const int* pts_cnt, int B, int M, int want_adi, float* out, void* stream) {
    if (B < 0 || M < 0) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!R_est || !t_est || !R_gt || !t_gt || !pts || !out) return fail(1, "null pointer");
    if ((pts_off == nullptr) != (pts_cnt == nullptr)) return fail(1, "pts_off and pts_cnt go together");
    lc::MetricsParams p{R_est, t_est, R_gt, t_gt, pts, pts_off, pts_cnt, out, B, M, want_adi};
    return lc::launch_pose_errors(p, static_cast<hipStream_t>(stream)) ? fail(11, "pose-error kernel launch failed") : 0;
}

int lc_sqnorm(const void* x, int dtype, long long n, double* partials, unsigned* ticket, float* sq, int accumulate, const float* state,
              float* state_snapshot, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(1, "dtype must be LC_F32, LC_F16 or LC_BF16");
    if (n < 0) return fail(1, "bad size");
    if (!partials || !ticket || !sq || (n > 0 && !x)) return fail(1, "null pointer");
    if ((state == nullptr) != (state_snapshot == nullptr)) return fail(1, "state and state_snapshot go together");
    lc::ClipParams p{};
    p.x = x; p.n = n; p.vec = !misaligned(dtype ? 8 : 16, x); p.partials = partials; p.ticket = ticket; p.sq = sq; p.accumulate = accumulate;
    p.state_in = state; p.state_snapshot = state_snapshot; p.dtype = dtype;
    return lc::launch_sqnorm(p, static_cast<hipStream_t>(stream)) ? fail(11, "sqnorm launch failed") : 0;
}

int lc_norm_clip_apply(const void* grad, int dtype, long long n, const float* sq, const float* state_in, float initial_max_norm, float scale,
                       double momentum, void* out, float* state_out, float* norm_out, void* stream) {
    if (dtype < 0 || dtype > 2) return fail(1, "dtype must be LC_F32, LC_F16 or LC_BF16");
    if (n < 0) return fail(1, "bad size");
    if (!sq || !state_in || (n > 0 && (!grad || !out))) return fail(1, "null pointer");
    lc::ClipParams p{};
    p.x = grad; p.n = n; p.vec = !misaligned(dtype ? 8 : 16, grad, out); p.sq = const_cast<float*>(sq); p.state_in = state_in;
    p.initial_max_norm = initial_max_norm; p.scale = scale; p.keep = (float)(1.0 - momentum); p.gain = (float)(momentum * (double)scale);
    p.out = out; p.state_out = state_out; p.norm_out = norm_out; p.dtype = dtype;
    return lc::launch_clip_apply(p, static_cast<hipStream_t>(stream)) ? fail(11, "clip launch failed") : 0;
}

int lc_kpt_nll_fwd_bwd_f32(const float* K, const float* pose, const float* pts3d, const float* pts2d, const float* pts2d_std, int B,
                           int N, float* nll, float* d_pts2d, float* d_std, void* stream) {
    if (B < 0 || N <= 0) return fail(1, "bad size");
    if (B == 0) return 0;
    if (!K || !pose || !pts3d || !pts2d || !pts2d_std || !nll) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, pts2d, pts2d_std, d_pts2d, d_std);
    lc::KptParams p{K, pose, pts3d, pts2d, pts2d_std, nll, d_pts2d, d_std, B, N};
    return lc::launch_kpt_nll(p, static_cast<hipStream_t>(stream)) ? fail(11, "keypoint NLL launch failed") : 0;
}

int lc_dense_select_f32(const float* pts2d, const float* inv_std, const float* pts3d, const unsigned char* mask,
                        const int* in_counts, const int* in_index, int B, int N, int mode, double quantile, int square_weights,
                        int min_count, unsigned seed, float* out_pts2d, float* out_weights, float* out_pts3d, int* out_index,
                        int* counts, void* stream) {
    if (B < 0 || N <= 0 || mode < 0 || mode > 2 || min_count < 0 || min_count > N) return fail(1, "bad size or mode");
    if (mode != 0 && !(quantile >= 0.0 && quantile <= 1.0)) return fail(1, "quantile outside [0,1]");
    if (B == 0) return 0;
    if (!pts2d || !inv_std || !pts3d || !out_pts2d || !out_weights || !out_pts3d || !counts) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, pts2d, inv_std, out_pts2d, out_weights);
    if (mode != 1 && !mask) return fail(1, "modes 0 (mask) and 2 (quantile_in_mask) need a mask");
    lc::SelectParams p{pts2d, inv_std, pts3d, mask, in_counts, in_index, out_pts2d, out_weights, out_pts3d, out_index, counts,
                       B, N, mode, (float)quantile, (float)(1.0 - quantile), square_weights, min_count, seed};
    const int rc = lc::launch_dense_select(p, static_cast<hipStream_t>(stream));
    if (rc == 3) return fail(1, "more than 32768 points per sample do not fit the LDS sort");
    return rc ? fail(11, "dense select launch failed") : 0;
}

size_t lc_dense_frontend_select_workspace_bytes(int B, int H, int W, int top, int left, int sample) {
    if (B <= 0 || H <= 0 || W <= 0 || sample <= 0 || top < 0 || left < 0 || top >= H || left >= W) return 0;
    return lc::dense_select_split_workspace_bytes(B, ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample));
}

int lc_dense_frontend_select3(const void* xyz, const void* wlogits, const void* wscale, const float* noc_scale, const void* vis_logits,
                                 float vis_thresh, int map_dtype, int xyz_dtype, int wscale_dtype, long long xyz_bstride, long long wlogits_bstride, long long vis_bstride, int B, int H, int W, int top, int left, int sample, int mode, double quantile,
                                 int square_weights, int min_count, unsigned seed, int pose_index_offset, float* out_pts2d, float* out_weights, float* out_pts3d,
                                 int* out_index, int* counts, void* workspace, size_t workspace_bytes, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (wscale_dtype < 0 || wscale_dtype > 2) return fail(1, "wscale_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (xyz_dtype != map_dtype && xyz_dtype != 0) return fail(1, "xyz_dtype must be map_dtype or LC_F32");
    if (xyz_bstride < 0 || wlogits_bstride < 0 || vis_bstride < 0) return fail(1, "negative batch stride");
    if ((xyz_bstride && xyz_bstride < 3ll * H * W) || (wlogits_bstride && wlogits_bstride < 2ll * H * W) || (vis_bstride && vis_bstride < 1ll * H * W)) return fail(1, "batch stride smaller than a sample");
    if (B < 0 || H <= 0 || W <= 0 || sample <= 0 || top < 0 || left < 0 || top >= H || left >= W) return fail(1, "bad size");
    const int N = ((H - top + sample - 1) / sample) * ((W - left + sample - 1) / sample);
    if (mode < 0 || mode > 2 || min_count < 0 || min_count > N) return fail(1, "bad size or mode");
    if (mode != 0 && !(quantile >= 0.0 && quantile <= 1.0)) return fail(1, "quantile outside [0,1]");
    if (N > 16384) return fail(1, "more than 16384 sampled pixels per object: use lc_dense_frontend_fwd3 + lc_dense_select_f32");
    if (B == 0) return 0;
    if (!wlogits || !wscale || !out_pts2d || !out_weights || !counts) return fail(1, "null pointer");
    if ((xyz == nullptr) != (out_pts3d == nullptr)) return fail(1, "xyz and out_pts3d go together (both NULL: the selection alone, lc_bits_decode_rows fills the points)");
    if (mode != 1 && !vis_logits) return fail(1, "modes 0 (mask) and 2 (quantile_in_mask) need the visibility logits");
    LC_REQUIRE_ALIGNED(8, out_pts2d, out_weights);
    lc::SelectParams p{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out_pts2d, out_weights, out_pts3d, out_index, counts,
                       B, N, mode, (float)quantile, (float)(1.0 - quantile), square_weights, min_count, seed, pose_index_offset};
    if (workspace) {  // several workgroups per object where the shape takes them (lc_kernels.h: dense_select_split_parts)
        const size_t need = lc::dense_select_split_workspace_bytes(B, N);
        if (need) {
            if (workspace_bytes < need) return fail(1, "workspace smaller than lc_dense_frontend_select_workspace_bytes(B, H, W, top, left, sample)");
            if (reinterpret_cast<uintptr_t>(workspace) & 127u) return fail(1, "workspace must be 128-byte aligned");
            p.split_ws = workspace;
        }
    }
    lc::DenseParams d{xyz, wlogits, wscale, noc_scale, nullptr, nullptr, nullptr, nullptr, B, H, W, N, top, left, sample, vis_logits, vis_thresh, nullptr, map_dtype, wscale_dtype, xyz_dtype,
                       xyz_bstride ? xyz_bstride : 3ll * H * W, wlogits_bstride ? wlogits_bstride : 2ll * H * W, vis_bstride ? vis_bstride : 1ll * H * W};
    return lc::launch_dense_frontend_select(p, d, static_cast<hipStream_t>(stream)) ? fail(11, "front end + select launch failed") : 0;
}

int lc_dense_aux_fwd2(const void* xyz, const unsigned char* msk_noc_u8, const float* msk_noc_f32, const float* noc_tgt,
                         const void* seg_logits, const float* msk_vis, const void* wlogits, int map_dtype, long long xyz_bstride, long long seg_bstride, long long wlogits_bstride, int B, int HW, int seg_type, float* losses,
                         double* partials, unsigned* ticket, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (xyz_bstride < 0 || seg_bstride < 0 || wlogits_bstride < 0) return fail(1, "negative batch stride");
    if ((xyz_bstride && xyz_bstride < 3ll * HW) || (seg_bstride && seg_bstride < HW) || (wlogits_bstride && wlogits_bstride < 2ll * HW)) return fail(1, "batch stride smaller than a sample");
    if (xyz_bstride >= (1ll << 31) / (B > 0 ? B : 1) || seg_bstride >= (1ll << 31) / (B > 0 ? B : 1) || wlogits_bstride >= (1ll << 31) / (B > 0 ? B : 1)) return fail(1, "maps of 2^31 elements or more");
    if (B < 0 || HW <= 0 || seg_type < 0 || seg_type > 1) return fail(1, "bad size or loss type");
    if ((long long)B * HW * 3 >= (1ll << 31)) return fail(1, "maps of 2^31 elements or more");
    if (B == 0) return 0;
    if (!seg_logits || !msk_vis || !losses || !partials || !ticket) return fail(1, "null pointer");
    if (xyz && (!noc_tgt || (msk_noc_u8 != nullptr) == (msk_noc_f32 != nullptr))) return fail(1, "xyz needs its target and exactly one mask form");
    LC_REQUIRE_ALIGNED(8, partials);
    lc::DenseAuxParams p{xyz, msk_noc_u8, msk_noc_f32, noc_tgt, seg_logits, msk_vis, wlogits, seg_type, losses, partials, ticket,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, HW, map_dtype,
                         xyz_bstride ? xyz_bstride : 3ll * HW, seg_bstride ? seg_bstride : 1ll * HW, wlogits_bstride ? wlogits_bstride : 2ll * HW};
    return lc::launch_dense_aux_fwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "dense aux loss launch failed") : 0;
}

int lc_dense_aux_bwd2(const void* xyz, const unsigned char* msk_noc_u8, const float* msk_noc_f32, const float* noc_tgt,
                         const void* seg_logits, const float* msk_vis, const void* wlogits, int map_dtype, long long xyz_bstride, long long seg_bstride, long long wlogits_bstride, int B, int HW, int seg_type,
                         const float* g_noc, const float* g_seg, const float* g_wseg, void* d_xyz, void* d_seg, void* d_wlogits,
                         void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (xyz_bstride < 0 || seg_bstride < 0 || wlogits_bstride < 0) return fail(1, "negative batch stride");
    if ((xyz_bstride && xyz_bstride < 3ll * HW) || (seg_bstride && seg_bstride < HW) || (wlogits_bstride && wlogits_bstride < 2ll * HW)) return fail(1, "batch stride smaller than a sample");
    if (xyz_bstride >= (1ll << 31) / (B > 0 ? B : 1) || seg_bstride >= (1ll << 31) / (B > 0 ? B : 1) || wlogits_bstride >= (1ll << 31) / (B > 0 ? B : 1)) return fail(1, "maps of 2^31 elements or more");
    if (B < 0 || HW <= 0 || seg_type < 0 || seg_type > 1) return fail(1, "bad size or loss type");
    if ((long long)B * HW * 3 >= (1ll << 31)) return fail(1, "maps of 2^31 elements or more");
    if (B == 0) return 0;
    if (!msk_vis || (d_seg && !seg_logits) || (d_wlogits && !wlogits)) return fail(1, "null pointer");
    if (d_xyz && (!xyz || !noc_tgt || (msk_noc_u8 != nullptr) == (msk_noc_f32 != nullptr))) return fail(1, "d_xyz needs xyz, its target and exactly one mask form");
    lc::DenseAuxParams p{xyz, msk_noc_u8, msk_noc_f32, noc_tgt, seg_logits, msk_vis, wlogits, seg_type, nullptr, nullptr, nullptr,
                         g_noc, g_seg, g_wseg, d_xyz, d_seg, d_wlogits, B, HW, map_dtype,
                         xyz_bstride ? xyz_bstride : 3ll * HW, seg_bstride ? seg_bstride : 1ll * HW, wlogits_bstride ? wlogits_bstride : 2ll * HW};
    return lc::launch_dense_aux_bwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "dense aux loss backward launch failed") : 0;
}

static int xyz_bin_fwd_common(const void* logits, const unsigned char* gt_bits, const void* msk_vis_logits, int map_dtype, long long logits_bstride, long long vis_bstride,
                              int B, int C, int HW, float momentum, float* histogram, float* loss, float* bin_weights, long long* counts, float* bce_mean,
                              double* partials, unsigned* ticket, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0 || vis_bstride < 0) return fail(1, "negative batch stride");
    if ((logits_bstride && logits_bstride < (long long)C * HW) || (vis_bstride && vis_bstride < HW)) return fail(1, "batch stride smaller than a sample");
    if (logits_bstride >= (1ll << 31) / (B > 0 ? B : 1) || vis_bstride >= (1ll << 31) / (B > 0 ? B : 1)) return fail(1, "maps of 2^31 elements or more");
    if (B < 0 || C <= 0 || HW <= 0) return fail(1, "bad size");
    if (C > lc::kBinMaxChannels) return fail(1, "more than 128 code bits");
    if ((long long)B * C * HW >= (1ll << 31)) return fail(1, "logits of 2^31 elements or more");
    if (B == 0) return 0;
    if (!logits || !gt_bits || !msk_vis_logits || !partials || !ticket) return fail(1, "null pointer");
    if (counts ? !bce_mean : (!histogram || !loss || !bin_weights)) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, partials);
    if (counts) LC_REQUIRE_ALIGNED(8, counts);
    const int vec = HW % 4 == 0 && ((logits_bstride | vis_bstride) & 3) == 0 && !misaligned(map_dtype ? 8 : 16, logits, msk_vis_logits) && !misaligned(4, gt_bits);
    lc::BinLossParams p{};
    p.logits = logits; p.gt_bits = gt_bits; p.msk_vis_logits = msk_vis_logits;
    p.histogram = histogram; p.momentum = momentum; p.loss = loss; p.bin_weights = bin_weights;
    p.partials = partials; p.ticket = ticket; p.counts_out = counts; p.bce_mean = bce_mean;
    p.B = B; p.C = C; p.HW = HW; p.vec = vec; p.map_dtype = map_dtype;
    p.logits_bs = logits_bstride ? logits_bstride : (long long)C * HW;
    p.vis_bs = vis_bstride ? vis_bstride : 1ll * HW;
    return lc::launch_xyz_bin_loss_fwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "code loss launch failed") : 0;
}

int lc_xyz_bin_loss_fwd2(const void* logits, const unsigned char* gt_bits, const void* msk_vis_logits, int map_dtype, long long logits_bstride, long long vis_bstride, int B, int C, int HW,
                            float momentum, float* histogram, float* loss, float* bin_weights, double* partials, unsigned* ticket,
                            void* stream) {
    return xyz_bin_fwd_common(logits, gt_bits, msk_vis_logits, map_dtype, logits_bstride, vis_bstride, B, C, HW, momentum, histogram, loss, bin_weights, nullptr, nullptr,
                              partials, ticket, stream);
}

int lc_xyz_bin_loss_counts(const void* logits, const unsigned char* gt_bits, const void* msk_vis_logits, int map_dtype, long long logits_bstride, long long vis_bstride, int B, int C,
                           int HW, long long* counts, float* bce_mean, double* partials, unsigned* ticket, void* stream) {
    if (!counts) return fail(1, "null pointer");
    return xyz_bin_fwd_common(logits, gt_bits, msk_vis_logits, map_dtype, logits_bstride, vis_bstride, B, C, HW, 0.f, nullptr, nullptr, nullptr, counts, bce_mean, partials,
                              ticket, stream);
}

int lc_xyz_bin_loss_finish(const long long* counts, const float* bce_mean, int C, float momentum, float* histogram, float* loss, float* bin_weights, void* stream) {
    if (C <= 0) return fail(1, "bad size");
    if (C > lc::kBinMaxChannels) return fail(1, "more than 128 code bits");
    if (!counts || !bce_mean || !histogram || !loss || !bin_weights) return fail(1, "null pointer");
    LC_REQUIRE_ALIGNED(8, counts);
    lc::BinLossParams p{};
    p.counts_in = counts; p.bce_mean = const_cast<float*>(bce_mean); p.C = C; p.momentum = momentum;
    p.histogram = histogram; p.loss = loss; p.bin_weights = bin_weights;
    return lc::launch_xyz_bin_loss_finish(p, static_cast<hipStream_t>(stream)) ? fail(11, "code loss finish launch failed") : 0;
}

int lc_xyz_bin_loss_bwd2(const void* logits, const unsigned char* gt_bits, const void* msk_vis_logits, const float* bin_weights,
                            const float* g_loss, int map_dtype, long long logits_bstride, long long vis_bstride, int B, int C, int HW, void* d_logits, void* stream) {
    if (map_dtype < 0 || map_dtype > 2) return fail(1, "map_dtype must be LC_F32, LC_F16 or LC_BF16");
    if (logits_bstride < 0 || vis_bstride < 0) return fail(1, "negative batch stride");
    if ((logits_bstride && logits_bstride < (long long)C * HW) || (vis_bstride && vis_bstride < HW)) return fail(1, "batch stride smaller than a sample");
    if (logits_bstride >= (1ll << 31) / (B > 0 ? B : 1) || vis_bstride >= (1ll << 31) / (B > 0 ? B : 1)) return fail(1, "maps of 2^31 elements or more");
    if (B < 0 || C <= 0 || HW <= 0) return fail(1, "bad size");
    if ((long long)B * C * HW >= (1ll << 31)) return fail(1, "logits of 2^31 elements or more");
    if (B == 0) return 0;
    if (!logits || !gt_bits || !msk_vis_logits || !bin_weights || !g_loss || !d_logits) return fail(1, "null pointer");
    const int vec = HW % 4 == 0 && ((logits_bstride | vis_bstride) & 3) == 0 && !misaligned(map_dtype ? 8 : 16, logits, msk_vis_logits, d_logits) && !misaligned(4, gt_bits);
    lc::BinLossParams p{};
    p.logits = logits; p.gt_bits = gt_bits; p.msk_vis_logits = msk_vis_logits; p.bin_weights = const_cast<float*>(bin_weights);
    p.g_loss = g_loss; p.d_logits = d_logits; p.B = B; p.C = C; p.HW = HW; p.vec = vec; p.map_dtype = map_dtype;
    p.logits_bs = logits_bstride ? logits_bstride : (long long)C * HW;
    p.vis_bs = vis_bstride ? vis_bstride : 1ll * HW;
    return lc::launch_xyz_bin_loss_bwd(p, static_cast<hipStream_t>(stream)) ? fail(11, "code loss backward launch failed") : 0;
}

}  // extern "C"
#pragma GCC visibility pop
