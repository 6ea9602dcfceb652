/* Sanitizer driver for the C oracle (CPU only): random problems incl. degenerate ones (too few points, points behind the
 * camera, zero weights, huge outliers) through pnp_oracle_batched_f32 under -fsanitize=address,undefined.
 * Built and run by tests/test_oracle_sanitizers.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void pnp_oracle_batched_f32(float *states, const float *Ks, const float *pts2d, const float *pts3d, const float *sqrtL,
                            const int *ptCnts, int nmax, int maxIterCnt, float function_tolerance, float *result_trs, int *rets,
                            int job_count, int num_threads);

static float rnd(void) { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

int main(void) {
    enum { B = 64, N = 40 };
    float *st = malloc(sizeof(float) * B * 7), *K = malloc(sizeof(float) * B * 9), *u = malloc(sizeof(float) * B * N * 2);
    float *X = malloc(sizeof(float) * B * N * 3), *L = malloc(sizeof(float) * B * N * 4), *tr = malloc(sizeof(float) * B);
    int *cnt = malloc(sizeof(int) * B), *ret = malloc(sizeof(int) * B);
    srand(7);
    for (int b = 0; b < B; ++b) {
        float *k = K + 9 * b;
        k[0] = 250; k[1] = 3 * rnd(); k[2] = 32; k[3] = 3 * rnd(); k[4] = 250; k[5] = 32; k[6] = 0; k[7] = 0; k[8] = 1;
        float q[4] = {1.f, 0.2f * rnd(), 0.2f * rnd(), 0.2f * rnd()};
        float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int i = 0; i < 4; ++i) q[i] /= nq;
        float t[3] = {10 * rnd(), 10 * rnd(), (b % 9 == 8) ? -300.f : 800.f}; /* some objects behind the camera */
        float R[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[3] * q[0]), 2 * (q[1] * q[3] + q[2] * q[0]),
                      2 * (q[1] * q[2] + q[3] * q[0]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[1] * q[0]),
                      2 * (q[1] * q[3] - q[2] * q[0]), 2 * (q[2] * q[3] + q[1] * q[0]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
        for (int n = 0; n < N; ++n) {
            float *x = X + 3 * (b * N + n);
            for (int d = 0; d < 3; ++d) x[d] = 40 * rnd();
            float c[3];
            for (int d = 0; d < 3; ++d) c[d] = R[3 * d] * x[0] + R[3 * d + 1] * x[1] + R[3 * d + 2] * x[2] + t[d];
            u[2 * (b * N + n)] = 250 * c[0] / c[2] + 32 + rnd() + ((n % 11 == 0) ? 500 * rnd() : 0);
            u[2 * (b * N + n) + 1] = 250 * c[1] / c[2] + 32 + rnd();
            float *l = L + 4 * (b * N + n);
            l[0] = (b % 7 == 3) ? 0.f : 1 + 0.5f * rnd(); /* some jobs with zero weights */
            l[1] = 123.f; /* element [0,1] is ignored by contract */
            l[2] = 0.3f * rnd();
            l[3] = (b % 7 == 3) ? 0.f : 1 + 0.5f * rnd();
        }
        for (int i = 0; i < 4; ++i) st[7 * b + i] = q[i] + 0.05f * rnd();
        for (int i = 0; i < 3; ++i) st[7 * b + 4 + i] = t[i] * (1 + 0.03f * rnd());
        cnt[b] = (b % 5 == 0) ? b % 4 : 3 + (b * 7) % (N - 2); /* 0..3 points and ragged counts */
    }
    for (int threads = 1; threads <= 4; threads += 3)
        for (int iters = 1; iters <= 50; iters += 49)
            pnp_oracle_batched_f32(st, K, u, X, L, cnt, N, iters, 1e-6f, tr, ret, B, threads);
    int bad = 0, invalid = 0;
    for (int b = 0; b < B; ++b) {
        invalid += ret[b];
        for (int i = 0; i < 7; ++i) bad += !isfinite(st[7 * b + i]);
        if (cnt[b] < 3 && (ret[b] != 1 || tr[b] != 1.f)) ++bad;
    }
    printf("oracle sanitize: %d jobs, %d invalid, %d contract violations\n", B, invalid, bad);
    free(st); free(K); free(u); free(X); free(L); free(tr); free(cnt); free(ret);
    return bad;
}
