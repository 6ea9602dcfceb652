// Standalone C-ABI harness (no torch, no Python): what a C/C++ maintainer of the reference links against.  Calls the entry
// points of liblc_amd.so on small synthetic inputs and CHECKS the answers (exit code = number of failed checks);
// tests/test_gpu_native_harness.py builds and runs it on the GPU box.
// Build: hipcc --offload-arch=gfx950 -O2 tests/native/harness.cpp -Llc_amd/_C -llc_amd -Wl,-rpath,$PWD/lc_amd/_C -o /tmp/harness
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/lc_amd.h"

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            std::exit(2);                                                       \
        }                                                                       \
    } while (0)

template <typename T>
T* to_dev(const std::vector<T>& v) {
    T* d;
    CK(hipMalloc(&d, v.size() * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static int g_failed = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                      \
        }                                                                    \
    } while (0)

int main(int argc, char** argv) {
    const int B = 4, N = 64;
    const char* which = argc > 1 ? argv[1] : "all";
    std::vector<float> K(B * 9), pose(B * 7), X(B * N * 3), u(B * N * 2), s(B * N * 2), bbox(B * 24), start(B * 7);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int b = 0; b < B; ++b) {
        float* k = &K[b * 9];
        k[0] = 250; k[1] = 0; k[2] = 32; k[3] = 0; k[4] = 250; k[5] = 32; k[6] = 0; k[7] = 0; k[8] = 1;
        float* p = &pose[b * 7];
        p[0] = 1; p[1] = 0.1f * rnd(); p[2] = 0.1f * rnd(); p[3] = 0.1f * rnd();
        float nq = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
        for (int i = 0; i < 4; ++i) p[i] /= nq;
        p[4] = 10 * rnd(); p[5] = 10 * rnd(); p[6] = 800;
        for (int i = 0; i < 7; ++i) start[b * 7 + i] = p[i];
        start[b * 7 + 6] *= 1.02f;
        for (int k8 = 0; k8 < 8; ++k8)
            for (int d = 0; d < 3; ++d) bbox[b * 24 + k8 * 3 + d] = 40.f * (((k8 >> (2 - d)) & 1) ? -1.f : 1.f);
        float q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        float R[9] = {1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 - q3 * q0), 2 * (q1 * q3 + q2 * q0),
                      2 * (q1 * q2 + q3 * q0), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 - q1 * q0),
                      2 * (q1 * q3 - q2 * q0), 2 * (q2 * q3 + q1 * q0), 1 - 2 * (q1 * q1 + q2 * q2)};
        for (int n = 0; n < N; ++n) {
            float* x = &X[(b * N + n) * 3];
            for (int d = 0; d < 3; ++d) x[d] = 40 * rnd();
            float c[3];
            for (int d = 0; d < 3; ++d) c[d] = R[3 * d] * x[0] + R[3 * d + 1] * x[1] + R[3 * d + 2] * x[2] + p[4 + d];
            u[(b * N + n) * 2] = 250 * c[0] / c[2] + 32 + rnd();
            u[(b * N + n) * 2 + 1] = 250 * c[1] / c[2] + 32 + rnd();
            s[(b * N + n) * 2] = 1 + 0.5f * rnd();
            s[(b * N + n) * 2 + 1] = 1 + 0.5f * rnd();
        }
    }
    float *dK = to_dev(K), *dP = to_dev(pose), *dX = to_dev(X), *dU = to_dev(u), *dS = to_dev(s), *dB = to_dev(bbox), *dSt = to_dev(start);
    std::printf("lc_amd version %d, test=%s\n", lc_amd_version(), which);
    bool all = std::string(which) == "all";
    if (all || std::string(which) == "head") {
        const int M = 8, H = 64, W = 64;
        std::vector<float> lg(M * H * W);
        for (auto& v : lg) v = rnd();
        lg[17 * 64 + 42] = 12;
        float* dl = to_dev(lg);
        float *mean, *sd, *st, *gm, *gs, *gi;
        CK(hipMalloc(&mean, M * 2 * 4)); CK(hipMalloc(&sd, M * 2 * 4)); CK(hipMalloc(&st, M * 4 * 4));
        CK(hipMalloc(&gi, M * H * W * 4));
        gm = to_dev(std::vector<float>(M * 2, 1.f)); gs = to_dev(std::vector<float>(M * 2, 1.f));
        int rc = lc_softargmax2d_fwd(dl, LC_F32, M, H, W, 0, mean, sd, st, nullptr);
        CK(hipDeviceSynchronize());
        int rc2 = lc_softargmax2d_bwd(dl, LC_F32, mean, sd, st, gm, gs, M, H, W, 0, gi, nullptr);
        CK(hipDeviceSynchronize());
        float hm[2], hs[2];
        CK(hipMemcpy(hm, mean, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, sd, 8, hipMemcpyDeviceToHost));
        std::printf("head rc=%d/%d mean=(%.3f,%.3f) std=(%.3f,%.3f)\n", rc, rc2, hm[0], hm[1], hs[0], hs[1]);
        EXPECT(rc == 0 && rc2 == 0);
        EXPECT(std::fabs(hm[0] - 42.f) < 2.f && std::fabs(hm[1] - 17.f) < 2.f);  // the e^12 bump at (x=42, y=17) dominates the map
    }
    if (all || std::string(which) == "pnp") {
        float* tr; int* ret; int* it;
        CK(hipMalloc(&tr, B * 4)); CK(hipMalloc(&ret, B * 4)); CK(hipMalloc(&it, B * 4));
        int rc = lc_pnp_lm3_f32(dK, dX, dU, nullptr, dS, nullptr, nullptr, nullptr, dSt, tr, ret, it, B, N, 50, 1e-6f, 0, 0, nullptr, 0, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> st(B * 7), htr(B); std::vector<int> hr(B), hi(B);
        CK(hipMemcpy(st.data(), dSt, B * 28, hipMemcpyDeviceToHost)); CK(hipMemcpy(htr.data(), tr, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), ret, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hi.data(), it, B * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            std::printf("pnp rc=%d job %d ret=%d iters=%d tr=%g  tz=%.4f (gt %.4f)\n", rc, b, hr[b], hi[b], htr[b], st[b * 7 + 6], pose[b * 7 + 6]);
        EXPECT(rc == 0);
        for (int b = 0; b < B; ++b) EXPECT(hr[b] == 0 && hi[b] >= 1 && std::fabs(st[b * 7 + 6] - pose[b * 7 + 6]) < 20.f);  // +-1 px of noise at f = 250, z = 800

        // the reference's own entry point: host arrays of pointers in, states updated in place (lib/pnp/cxx/ext.h:2-15)
        std::vector<float> hst(start), L(B * N * 4, 0.f), rtr(B, -1.f);
        for (int i = 0; i < B * N; ++i) { L[4 * i] = s[2 * i]; L[4 * i + 3] = s[2 * i + 1]; }
        std::vector<float*> pS(B), pK(B), pU(B), pX(B), pL(B);
        std::vector<int> cnt(B, N), flags(B, -1);
        for (int b = 0; b < B; ++b) { pS[b] = &hst[b * 7]; pK[b] = &K[b * 9]; pU[b] = &u[b * N * 2]; pX[b] = &X[b * N * 3]; pL[b] = &L[b * N * 4]; }
        cnt[B - 1] = 2;  // too few points: flagged, state untouched, trust region 1 (ceres.cpp:84-91)
        pnp_ceres_f32_omp(pS.data(), pK.data(), pU.data(), pX.data(), pL.data(), cnt.data(), 50, 1e-6f, 0, rtr.data(), flags.data(), B, 2);
        for (int b = 0; b < B - 1; ++b) {
            EXPECT(flags[b] == 0);
            for (int i = 0; i < 7; ++i) EXPECT(std::fabs(hst[b * 7 + i] - st[b * 7 + i]) < 1e-4f * (i < 4 ? 1.f : 1000.f));  // same as the device route
        }
        EXPECT(flags[B - 1] == 1 && rtr[B - 1] == 1.f && hst[(B - 1) * 7 + 6] == start[(B - 1) * 7 + 6]);
        std::printf("host abi: flags %d %d %d %d\n", flags[0], flags[1], flags[2], flags[3]);
    }
    if (all || std::string(which) == "loss") {
        float *loss, *du, *ds, *dx, *aux;
        CK(hipMalloc(&loss, B * 4)); CK(hipMalloc(&du, B * N * 8)); CK(hipMalloc(&ds, B * N * 8)); CK(hipMalloc(&dx, B * N * 12));
        CK(hipMalloc(&aux, B * 40 * 4));
        int rc = lc_cov_loss3_fwd_bwd_f32(dK, dP, dX, dU, dS, nullptr, dB, nullptr, B, N, 32.f, 3.f, 4.f, 0, loss, du, ds, dx, aux, nullptr, 0, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> hl(B), hdu(4);
        CK(hipMemcpy(hl.data(), loss, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdu.data(), du, 16, hipMemcpyDeviceToHost));
        std::printf("loss rc=%d loss=(%.5f %.5f %.5f %.5f) du0=(%g %g)\n", rc, hl[0], hl[1], hl[2], hl[3], hdu[0], hdu[1]);
        EXPECT(rc == 0);
        for (int b = 0; b < B; ++b) EXPECT(std::isfinite(hl[b]) && hl[b] > -20.f && hl[b] < 20.f);
        EXPECT(std::isfinite(hdu[0]) && hdu[0] != 0.f);
        // error path: an 8-byte row pointer at a 4-byte offset is refused with a message, nothing is launched
        rc = lc_cov_loss3_fwd_bwd_f32(dK, dP, dX, dU + 1, dS, nullptr, dB, nullptr, B, N - 1, 32.f, 3.f, 4.f, 0, loss, du, ds, dx, aux, nullptr, 0, nullptr);
        EXPECT(rc != 0 && std::string(lc_amd_last_error()).find("aligned") != std::string::npos);
    }
    if (all || std::string(which) == "glue") {
        // keypoint NLL (losses.py:318-326): per-sample sums + gradients
        float *nll, *gu, *gs2;
        CK(hipMalloc(&nll, B * 4)); CK(hipMalloc(&gu, B * N * 8)); CK(hipMalloc(&gs2, B * N * 8));
        int rc = lc_kpt_nll_fwd_bwd_f32(dK, dP, dX, dU, dS, B, N, nll, gu, gs2, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> hn(B);
        CK(hipMemcpy(hn.data(), nll, B * 4, hipMemcpyDeviceToHost));
        EXPECT(rc == 0);
        for (int b = 0; b < B; ++b) EXPECT(std::isfinite(hn[b]));
        // gradient clipping (grad.py:5-83): first call clips to initial_max_norm and starts the running maximum
        const long long n = (long long)B * N * 2;
        double* partials; unsigned* ticket; float *sq, *state, *state2, *norm, *out;
        CK(hipMalloc(&partials, LC_SQNORM_BLOCKS * 8)); CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4));
        CK(hipMalloc(&sq, 4)); CK(hipMalloc(&state2, 4)); CK(hipMalloc(&norm, 4)); CK(hipMalloc(&out, n * 4));
        state = to_dev(std::vector<float>(1, -1.f));
        rc = lc_sqnorm(dU, LC_F32, n, partials, ticket, sq, 0, nullptr, nullptr, nullptr);
        int rc2 = lc_norm_clip_apply(dU, LC_F32, n, sq, state, 100.f, 1.7f, 0.1, out, state2, norm, nullptr);
        CK(hipDeviceSynchronize());
        float hsq, hs2, hnorm, ho;
        CK(hipMemcpy(&hsq, sq, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs2, state2, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hnorm, norm, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ho, out, 4, hipMemcpyDeviceToHost));
        double ref = 0;
        for (float v : u) ref += (double)v * v;
        EXPECT(rc == 0 && rc2 == 0);
        EXPECT(std::fabs(hsq - ref) < 1e-5 * ref && std::fabs(hnorm - std::sqrt(ref)) < 1e-5 * std::sqrt(ref));
        EXPECT(std::fabs(hs2 - 1.7f * hnorm) < 1e-4f * hs2);
        EXPECT(std::fabs(ho - u[0] * std::fmin(100.0 / (std::sqrt(ref) + 1e-6), 1.0)) < 1e-5f * std::fabs(u[0]) + 1e-7f);
        std::printf("glue: nll0=%.4f norm=%.3f max_norm=%.3f\n", hn[0], hnorm, hs2);
        // dense point selection (test.py:39-45,94-113): median split keeps half of the points, in source order
        float *ou, *ow, *ox; int *oi, *oc;
        CK(hipMalloc(&ou, B * N * 8)); CK(hipMalloc(&ow, B * N * 8)); CK(hipMalloc(&ox, B * N * 12)); CK(hipMalloc(&oi, B * N * 4)); CK(hipMalloc(&oc, B * 4));
        rc = lc_dense_select_f32(dU, dS, dX, nullptr, nullptr, nullptr, B, N, 1, 0.5, 1, 4, 0u, ou, ow, ox, oi, oc, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<int> hc(B), hidx(N);
        CK(hipMemcpy(hc.data(), oc, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hidx.data(), oi, N * 4, hipMemcpyDeviceToHost));
        EXPECT(rc == 0);
        for (int b = 0; b < B; ++b) EXPECT(hc[b] == N / 2);
        for (int i = 1; i < hc[0]; ++i) EXPECT(hidx[i] > hidx[i - 1]);
        std::printf("select: counts %d %d %d %d\n", hc[0], hc[1], hc[2], hc[3]);
    }
    if (all || std::string(which) == "split") {
        // Few poses x thousands of correspondences (the test-time solves behind the dense heads): what a C caller does for lc_pnp_lm3_f32 --
        // ask for the workspace size, zero the workspace ONCE, keep it for the following calls on the stream -- and what it gets: the
        // one-workgroup solve's poses (up to the order of the fp64 sums), flags 0 / 1 only, the same answer call after call (include/lc_amd.h)
        const int Bs = 8, Ns = 2304;
        std::vector<float> K2(Bs * 9), X2(Bs * Ns * 3), U2(Bs * Ns * 2), S2(Bs * Ns * 2), st0(Bs * 7);
        for (int b = 0; b < Bs; ++b) {
            for (int i = 0; i < 9; ++i) K2[b * 9 + i] = K[i];
            const float* p = &pose[(b % B) * 7];
            for (int i = 0; i < 7; ++i) st0[b * 7 + i] = p[i];
            st0[b * 7 + 6] *= 1.03f;
            float q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float R[9] = {1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 - q3 * q0), 2 * (q1 * q3 + q2 * q0),
                          2 * (q1 * q2 + q3 * q0), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 - q1 * q0),
                          2 * (q1 * q3 - q2 * q0), 2 * (q2 * q3 + q1 * q0), 1 - 2 * (q1 * q1 + q2 * q2)};
            for (int n = 0; n < Ns; ++n) {
                float* x = &X2[((size_t)b * Ns + n) * 3];
                for (int d = 0; d < 3; ++d) x[d] = 40 * rnd();
                float c[3];
                for (int d = 0; d < 3; ++d) c[d] = R[3 * d] * x[0] + R[3 * d + 1] * x[1] + R[3 * d + 2] * x[2] + p[4 + d];
                U2[((size_t)b * Ns + n) * 2] = 250 * c[0] / c[2] + 32 + rnd();
                U2[((size_t)b * Ns + n) * 2 + 1] = 250 * c[1] / c[2] + 32 + rnd();
                S2[((size_t)b * Ns + n) * 2] = 1 + 0.5f * rnd();
                S2[((size_t)b * Ns + n) * 2 + 1] = 1 + 0.5f * rnd();
            }
        }
        float *dK2 = to_dev(K2), *dX2 = to_dev(X2), *dU2 = to_dev(U2), *dS2 = to_dev(S2), *dSt0 = to_dev(st0);
        float *stA, *stB, *trA, *trB; int *retA, *retB, *itA, *itB;
        CK(hipMalloc(&stA, Bs * 28)); CK(hipMalloc(&stB, Bs * 28)); CK(hipMalloc(&trA, Bs * 4)); CK(hipMalloc(&trB, Bs * 4));
        CK(hipMalloc(&retA, Bs * 4)); CK(hipMalloc(&retB, Bs * 4)); CK(hipMalloc(&itA, Bs * 4)); CK(hipMalloc(&itB, Bs * 4));
        const size_t need = lc_pnp_lm_workspace_bytes(Bs, Ns);
        EXPECT(need > 0 && lc_pnp_lm_workspace_bytes(Bs, 64) == 0);  // this shape takes several workgroups per pose; the metric's shape does not
        void* ws = nullptr;
        CK(hipMalloc(&ws, need ? need : 128));
        CK(hipMemset(ws, 0, need ? need : 128));  // once
        int rc1 = lc_pnp_lm3_f32(dK2, dX2, dU2, nullptr, dS2, nullptr, nullptr, dSt0, stA, trA, retA, itA, Bs, Ns, 50, 1e-6f, 0, 0, nullptr, 0, nullptr);
        std::vector<float> a(Bs * 7), b2(Bs * 7); std::vector<int> ra(Bs), rb(Bs), ia(Bs), ib(Bs);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), stA, Bs * 28, hipMemcpyDeviceToHost)); CK(hipMemcpy(ra.data(), retA, Bs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ia.data(), itA, Bs * 4, hipMemcpyDeviceToHost));
        EXPECT(rc1 == 0);
        for (int rep = 0; rep < 3; ++rep) {
            int rc = lc_pnp_lm3_f32(dK2, dX2, dU2, nullptr, dS2, nullptr, nullptr, dSt0, stB, trB, retB, itB, Bs, Ns, 50, 1e-6f, 0, 0, ws, need, nullptr);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(b2.data(), stB, Bs * 28, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), retB, Bs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ib.data(), itB, Bs * 4, hipMemcpyDeviceToHost));
            EXPECT(rc == 0);
            for (int b = 0; b < Bs; ++b) {
                EXPECT(rb[b] == ra[b] && rb[b] == 0 && ib[b] == ia[b]);
                for (int i = 0; i < 7; ++i) EXPECT(std::fabs(b2[b * 7 + i] - a[b * 7 + i]) <= 2e-6f * (i < 4 ? 1.f : 1000.f));
            }
        }
        // a workspace that is too small or misaligned is refused with a message, nothing is launched
        EXPECT(lc_pnp_lm3_f32(dK2, dX2, dU2, nullptr, dS2, nullptr, nullptr, dSt0, stB, trB, retB, itB, Bs, Ns, 50, 1e-6f, 0, 0, ws, need - 1, nullptr) != 0);
        EXPECT(lc_pnp_lm3_f32(dK2, dX2, dU2, nullptr, dS2, nullptr, nullptr, dSt0, stB, trB, retB, itB, Bs, Ns, 50, 1e-6f, 0, 0, (char*)ws + 8, need, nullptr) != 0);
        std::printf("split: %zu workspace bytes for %d poses x %d points, iterations %d %d %d ...\n", need, Bs, Ns, ib[0], ib[1], ib[2]);
    }
    std::printf("harness done, %d check(s) failed\n", g_failed);
    return g_failed;
}
