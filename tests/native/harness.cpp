// Standalone C-ABI harness (no torch): exercises every entry point of liblc_amd.so with small synthetic inputs.
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
        int rc = lc_softargmax2d_fwd_f32(dl, M, H, W, 0, mean, sd, st, nullptr);
        CK(hipDeviceSynchronize());
        int rc2 = lc_softargmax2d_bwd_f32(dl, mean, sd, st, gm, gs, M, H, W, 0, gi, nullptr);
        CK(hipDeviceSynchronize());
        float hm[2], hs[2];
        CK(hipMemcpy(hm, mean, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, sd, 8, hipMemcpyDeviceToHost));
        std::printf("head rc=%d/%d mean=(%.3f,%.3f) std=(%.3f,%.3f)\n", rc, rc2, hm[0], hm[1], hs[0], hs[1]);
    }
    if (all || std::string(which) == "pnp") {
        float* tr; int* ret; int* it;
        CK(hipMalloc(&tr, B * 4)); CK(hipMalloc(&ret, B * 4)); CK(hipMalloc(&it, B * 4));
        int rc = lc_pnp_lm_f32(dK, dX, dU, nullptr, dS, nullptr, nullptr, dSt, tr, ret, it, B, N, 50, 1e-6f, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> st(B * 7), htr(B); std::vector<int> hr(B), hi(B);
        CK(hipMemcpy(st.data(), dSt, B * 28, hipMemcpyDeviceToHost)); CK(hipMemcpy(htr.data(), tr, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), ret, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hi.data(), it, B * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            std::printf("pnp rc=%d job %d ret=%d iters=%d tr=%g  tz=%.4f (gt %.4f)\n", rc, b, hr[b], hi[b], htr[b], st[b * 7 + 6], pose[b * 7 + 6]);
    }
    if (all || std::string(which) == "loss") {
        float *loss, *du, *ds, *dx, *aux;
        CK(hipMalloc(&loss, B * 4)); CK(hipMalloc(&du, B * N * 8)); CK(hipMalloc(&ds, B * N * 8)); CK(hipMalloc(&dx, B * N * 12));
        CK(hipMalloc(&aux, B * 40 * 4));
        int rc = lc_cov_loss_fwd_bwd_f32(dK, dP, dX, dU, dS, nullptr, dB, nullptr, B, N, 32.f, 3.f, 4.f, loss, du, ds, dx, aux, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> hl(B), hdu(4);
        CK(hipMemcpy(hl.data(), loss, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdu.data(), du, 16, hipMemcpyDeviceToHost));
        std::printf("loss rc=%d loss=(%.5f %.5f %.5f %.5f) du0=(%g %g)\n", rc, hl[0], hl[1], hl[2], hl[3], hdu[0], hdu[1]);
    }
    std::printf("harness done\n");
    return 0;
}
