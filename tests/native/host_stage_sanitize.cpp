// ASan/UBSan driver for lc_amd/csrc/lc_host_stage.h (CPU only; built and run by tests/test_host_stage_sanitizers.py).
// Every caller array is an exact-sized heap allocation, so one float read or written past what the contract allows
// (7 state floats, 6 K floats, ptCnt x {2,3,4} point floats; nothing of the point arrays of a job with ptCnt <= 0)
// trips AddressSanitizer.  The GPU kernel is replaced by a stand-in that writes the output sections the way the
// kernel does (states for converged jobs, radius, flag), so gather -> "solve" -> scatter is exercised end to end.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lc_amd/csrc/lc_host_stage.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed line %d: %s\n", __LINE__, #c); ++failures; } } while (0)

static float rnd() { return (float)std::rand() / (float)RAND_MAX * 2.f - 1.f; }

static void run_case(const std::vector<int>& counts, const char* what) {
    const int B = (int)counts.size();
    std::vector<float*> st(B), K(B), u(B), X(B), L(B);
    std::vector<std::vector<float>> st0(B);
    for (int i = 0; i < B; ++i) {
        const int n = counts[i] > 0 ? counts[i] : 0;
        st[i] = (float*)std::malloc(7 * sizeof(float));
        K[i] = (float*)std::malloc(6 * sizeof(float));  // only 6 floats are guaranteed readable (ceres.cpp:99-101)
        // jobs without points hand over null pointers: nothing of them may be touched
        u[i] = n ? (float*)std::malloc(sizeof(float) * 2 * n) : nullptr;
        X[i] = n ? (float*)std::malloc(sizeof(float) * 3 * n) : nullptr;
        L[i] = n ? (float*)std::malloc(sizeof(float) * 4 * n) : nullptr;
        for (int k = 0; k < 7; ++k) st[i][k] = rnd();
        for (int k = 0; k < 6; ++k) K[i][k] = 100 * rnd();
        for (int k = 0; k < 2 * n; ++k) u[i][k] = 64 * rnd();
        for (int k = 0; k < 3 * n; ++k) X[i][k] = 40 * rnd();
        for (int k = 0; k < 4 * n; ++k) L[i][k] = rnd();
        st0[i].assign(st[i], st[i] + 7);
    }
    const int pmax = lc::host::stage_max_points(counts.data(), B);
    const size_t P = (size_t)pmax;
    size_t off[lc::host::kStageSections];
    const size_t bytes = lc::host::stage_layout((size_t)B, P, off);
    char* h = (char*)std::malloc(bytes);  // exact size: a write past the layout trips ASan
    lc::host::stage_gather(h, off, P, st.data(), K.data(), u.data(), X.data(), L.data(), counts.data(), B);
    // what the kernel would see
    const float* hK = (const float*)(h + off[0]);
    const float* hX = (const float*)(h + off[1]);
    const float* hU = (const float*)(h + off[2]);
    const float* hL = (const float*)(h + off[3]);
    const int* hC = (const int*)(h + off[4]);
    float* hS = (float*)(h + off[5]);
    float* hT = (float*)(h + off[6]);
    int* hR = (int*)(h + off[7]);
    for (int i = 0; i < B; ++i) {
        const int n = counts[i] > 0 ? counts[i] : 0;
        CHECK(hC[i] == counts[i]);
        CHECK(hK[9 * i + 8] == 1.f && hK[9 * i + 6] == 0.f);
        for (int k = 0; k < 6; ++k) CHECK(hK[9 * i + k] == K[i][k]);
        for (int k = 0; k < 3 * n; ++k) CHECK(hX[3 * P * i + k] == X[i][k]);
        for (size_t k = 3 * (size_t)n; k < 3 * P; ++k) CHECK(hX[3 * P * i + k] == 0.f);  // zero padding (cer_solver.py:67-87)
        for (int k = 0; k < 2 * n; ++k) CHECK(hU[2 * P * i + k] == u[i][k]);
        for (size_t k = 2 * (size_t)n; k < 2 * P; ++k) CHECK(hU[2 * P * i + k] == 0.f);
        for (int k = 0; k < 4 * n; ++k) CHECK(hL[4 * P * i + k] == L[i][k]);
        for (size_t k = 4 * (size_t)n; k < 4 * P; ++k) CHECK(hL[4 * P * i + k] == 0.f);
        // stand-in for the kernel: < 3 points -> invalid, radius 1; every third remaining job "does not converge"
        const bool invalid = counts[i] < 3 || (i % 3) == 2;
        hR[i] = invalid ? 1 : 0;
        hT[i] = counts[i] < 3 ? 1.f : 1e4f + i;
        for (int k = 0; k < 7; ++k) hS[7 * i + k] = invalid ? -777.f : (float)(i * 10 + k);  // garbage in invalid rows must not leak out
    }
    std::vector<float> tr(B);
    std::vector<int> ret(B);
    lc::host::stage_scatter(h, off, st.data(), tr.data(), ret.data(), B);
    for (int i = 0; i < B; ++i) {
        const bool invalid = counts[i] < 3 || (i % 3) == 2;
        CHECK(ret[i] == (invalid ? 1 : 0));
        CHECK(tr[i] == (counts[i] < 3 ? 1.f : 1e4f + i));
        for (int k = 0; k < 7; ++k) CHECK(st[i][k] == (invalid ? st0[i][k] : (float)(i * 10 + k)));  // in place only on success
    }
    std::free(h);
    for (int i = 0; i < B; ++i) { std::free(st[i]); std::free(K[i]); std::free(u[i]); std::free(X[i]); std::free(L[i]); }
    std::printf("case %-28s B=%d pmax=%d staging=%zu bytes ok\n", what, B, pmax, bytes);
}

int main() {
    std::srand(11);
    run_case({48, 40, 2, 17, 48, 3, 48, 48, 0, 48, 31, 48}, "ragged");
    run_case({0, 0, 0}, "all empty");
    run_case({-5, 3, -1, 7}, "negative counts");
    run_case({1}, "single tiny job");
    run_case({64}, "single full job");
    {  // > 4096 x 64 points: the size beyond which lc_capi.hip switches from the zero-copy route to explicit copies
        std::vector<int> big(4200);
        for (size_t i = 0; i < big.size(); ++i) big[i] = (i % 97 == 0) ? 0 : 1 + (int)(std::rand() % 80);
        big[17] = 80;
        run_case(big, "4200 jobs x up to 80 points");
    }
    std::printf("%d contract violations\n", failures);
    return failures ? 1 : 0;
}
