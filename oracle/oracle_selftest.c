/*
 * oracle/oracle_selftest.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Runs the CPU restatement (pcg_oracle.c) under AddressSanitizer / UBSan on the CPU
 * (`make -C oracle asan`; tests/test_oracle.py runs the binary): the input system of
 * /root/reference/examples/pcg_solve_dp.cu:14-25 and a batch of ragged small shapes with poisoned
 * corner blocks (L_0, R_{N-1} are never read, pcg.cuh:105-106), every flag combination, both
 * precisions, 1 and 3 OpenMP threads.  Prints one line per case; exit code 0 = no sanitizer report
 * (the sanitizers abort with their own message otherwise) and the known iteration counts hold.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pcg_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0 - 0.5;
}

/* diagonally dominant symmetric block-tridiagonal system, exact-size buffers (so that an
 * out-of-bounds access lands in a redzone) */
static int ragged_case(uint32_t n, uint32_t N, uint32_t batch, int flags, int threads)
{
    const size_t ms = (size_t)3 * n * n * N, vs = (size_t)n * N;
    double *S = malloc(sizeof(double) * ms * batch), *g = malloc(sizeof(double) * vs * batch);
    double *lam = calloc(vs * batch, sizeof(double)), *r = malloc(sizeof(double) * vs * batch);
    double *p = malloc(sizeof(double) * vs * batch), *y = malloc(sizeof(double) * vs * batch);
    float *S32 = malloc(sizeof(float) * ms * batch), *g32 = malloc(sizeof(float) * vs * batch);
    float *lam32 = calloc(vs * batch, sizeof(float));
    uint32_t *it = malloc(sizeof(uint32_t) * batch);
    uint8_t *ex = malloc(batch);
    int bad = 0;
    for (uint32_t b = 0; b < batch; ++b) {
        double *M = S + b * ms;
        for (size_t i = 0; i < ms; ++i) M[i] = 0.0;
        for (uint32_t k = 0; k < N; ++k) {
            for (uint32_t c = 0; c < n; ++c)
                for (uint32_t rr = 0; rr <= c; ++rr) {
                    const double v = rr == c ? 4.0 + 3.0 * n : 0.3 * rnd();
                    M[(size_t)k * 3 * n * n + (size_t)n * n + c * n + rr] = v; /* D_k symmetric */
                    M[(size_t)k * 3 * n * n + (size_t)n * n + rr * n + c] = v;
                }
            if (k + 1 < N)
                for (uint32_t c = 0; c < n; ++c)
                    for (uint32_t rr = 0; rr < n; ++rr) {
                        const double v = 0.3 * rnd();
                        M[(size_t)k * 3 * n * n + (size_t)2 * n * n + c * n + rr] = v;       /* R_k      */
                        M[(size_t)(k + 1) * 3 * n * n + (size_t)rr * n + c] = v;             /* L_{k+1} = R_k^T */
                    }
        }
        for (uint32_t i = 0; i < n * n; ++i) {   /* never read: poison */
            M[i] = NAN;
            M[(size_t)(N - 1) * 3 * n * n + (size_t)2 * n * n + i] = NAN;
        }
        for (size_t i = 0; i < vs; ++i) g[b * vs + i] = rnd();
    }
    for (size_t i = 0; i < ms * batch; ++i) S32[i] = (float)S[i];
    for (size_t i = 0; i < vs * batch; ++i) g32[i] = (float)g[i];
    bad |= oracle_pcg_batch_f64(n, N, batch, S, NULL, g, lam, r, p, 1e-12, 200, it, ex, flags, threads);
    for (uint32_t b = 0; b < batch; ++b) bad |= ex[b] != 0 || it[b] == 0;
    bad |= oracle_spmv_batch_f64(n, N, batch, S, lam, y, flags, threads);
    double worst = 0.0;
    for (size_t i = 0; i < vs * batch; ++i) {
        const double e = fabs(y[i] - g[i]);
        if (!(e <= worst)) worst = e;   /* NaN-propagating max */
    }
    bad |= !(worst < 1e-5);
    bad |= oracle_pcg_batch_f32(n, N, batch, S32, NULL, g32, lam32, NULL, NULL, 1e-6f, 200, it, ex, flags, threads);
    bad |= oracle_pcg_f32(n, N, S32, NULL, g32, lam32, NULL, NULL, 1e-6f, 3, it, ex, NULL, flags);
    printf("ragged n=%u N=%u batch=%u flags=%d threads=%d residual=%.3g %s\n", n, N, batch, flags, threads, worst,
           bad ? "FAIL" : "ok");
    free(S); free(g); free(lam); free(r); free(p); free(y); free(S32); free(g32); free(lam32); free(it); free(ex);
    return bad;
}

int main(void)
{
    int bad = 0;
    /* the reference's example system, examples/pcg_solve_dp.cu:14-25 (n = 2, N = 3) */
    const double S[36] = {0,     0,     0,     0,      -.999,  0,     0,     -.999,   .999, .0999, -.98, .999,
                          .999,  -.98,  .0999, .999,   -2.008, .8801, .8801, -3.0584, .999, .0999, -.98, .999,
                          .999,  -.98,  .0999, .999,   -1.019, .8801, .8801, -2.0694, 0,    0,     0,    0};
    const double gamma[6] = {3.1385, 0, 0, 3.0788, .0031, 3.0788};
    for (int flags = 0; flags < 4; ++flags) {
        double lam[6] = {0}, r[6], p[6], eta[26];
        uint32_t it = 0;
        uint8_t ex = 1;
        bad |= oracle_pcg_f64(2, 3, S, NULL, gamma, lam, r, p, 1e-6, 25, &it, &ex, eta, flags);
        printf("readme f64 flags=%d iters=%u exit=%u lambda0=%.9f\n", flags, it, (unsigned)ex, lam[0]);
        bad |= it != 6 || ex != 0 || fabs(lam[0] + 303.702986086) > 1e-6;   /* SURVEY.md section 8c (2) */
    }
    const uint32_t shapes[][3] = {{1, 1, 2}, {2, 3, 1}, {3, 1, 2}, {5, 2, 3}, {7, 9, 2}, {14, 4, 2}};
    for (unsigned s = 0; s < sizeof shapes / sizeof shapes[0]; ++s)
        for (int flags = 0; flags < 4; ++flags)
            bad |= ragged_case(shapes[s][0], shapes[s][1], shapes[s][2], flags, flags == 3 ? 3 : 1);
    printf(bad ? "oracle selftest FAILED\n" : "oracle selftest ok\n");
    return bad ? 1 : 0;
}
