// Exercises every entry of the C++ drop-in surface (include/gbdpcg.hpp) on the example system and
// prints one parseable line per call:  <tag> iters=<k> lambda=<6 values>.
// tests/test_gpu_cpp_api.py runs it on the GPU box and compares with the CPU oracle.
#include <cstdio>
#include <vector>

#include "gpu_pcg.cuh"

template <typename T> static void report(const char *tag, uint32_t iters, const T *lam)
{
    printf("%s iters=%u lambda=", tag, iters);
    for (int i = 0; i < 6; ++i) printf("%.17g ", (double)lam[i]);
    printf("\n");
}

template <typename T> static void run(const char *prec)
{
    const uint32_t n = 2, N = 3;
    T h_S[36] = {0,     0,     0,     0,      -.999,  0,     0,     -.999,   .999, .0999, -.98, .999,
                 .999,  -.98,  .0999, .999,   -2.008, .8801, .8801, -3.0584, .999, .0999, -.98, .999,
                 .999,  -.98,  .0999, .999,   -1.019, .8801, .8801, -2.0694, 0,    0,     0,    0};
    T h_gamma[6] = {3.1385, 0, 0, 3.0788, .0031, 3.0788};
    char tag[64];

    printf("%s smem=%zu occupancy=%d\n", prec, pcgSharedMemSize<T>(n, N), (int)checkPcgOccupancy<T>(nullptr, dim3(64), n, N));

    {   // host overload, identity preconditioner
        T lam[6] = {0};
        pcg_config<T> cfg;
        uint32_t it = solvePCG<T>(h_S, h_gamma, lam, n, N, &cfg);
        snprintf(tag, sizeof tag, "%s host_ident", prec);
        report(tag, it, lam);
    }
    {   // host overload, stair preconditioner formed on the device
        T lam[6] = {0};
        pcg_config<T> cfg;
        cfg.empty_pinv = 0;
        uint32_t it = solvePCG<T>(h_S, h_gamma, lam, n, N, &cfg);
        snprintf(tag, sizeof tag, "%s host_stair", prec);
        report(tag, it, lam);
    }
    {   // README spelling
        T lam[6] = {0};
        int it = pcg_solve<T>(h_S, h_gamma, lam, n, N);
        snprintf(tag, sizeof tag, "%s pcg_solve", prec);
        report(tag, (uint32_t)it, lam);
    }
    {   // device overload: caller owns every buffer, scratch included
        T *d_S, *d_P, *d_g, *d_l, *d_r, *d_p, *d_v, *d_e;
        gpuErrchk(hipMalloc((void **)&d_S, sizeof h_S));
        gpuErrchk(hipMalloc((void **)&d_P, sizeof h_S));
        gpuErrchk(hipMalloc((void **)&d_g, sizeof h_gamma));
        gpuErrchk(hipMalloc((void **)&d_l, sizeof h_gamma));
        gpuErrchk(hipMalloc((void **)&d_r, sizeof h_gamma));
        gpuErrchk(hipMalloc((void **)&d_p, sizeof h_gamma));
        gpuErrchk(hipMalloc((void **)&d_v, N * sizeof(T)));
        gpuErrchk(hipMalloc((void **)&d_e, N * sizeof(T)));
        gpuErrchk(hipMemcpy(d_S, h_S, sizeof h_S, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_g, h_gamma, sizeof h_gamma, hipMemcpyHostToDevice));
        gpuErrchk(hipMemset(d_l, 0, sizeof h_gamma));
        // identity preconditioner written out explicitly in the [L|D|R] layout
        std::vector<T> P(36, T(0));
        for (uint32_t k = 0; k < N; ++k)
            for (uint32_t i = 0; i < n; ++i) P[k * 12 + 4 + i * n + i] = T(1);
        gpuErrchk(hipMemcpy(d_P, P.data(), sizeof h_S, hipMemcpyHostToDevice));
        pcg_config<T> cfg;
        uint32_t it = solvePCG<T>(n, N, d_S, d_P, d_g, d_l, d_r, d_p, d_v, d_e, &cfg);
        T lam[6];
        gpuErrchk(hipMemcpy(lam, d_l, sizeof lam, hipMemcpyDeviceToHost));
        snprintf(tag, sizeof tag, "%s device_ident", prec);
        report(tag, it, lam);
        T r[6];
        gpuErrchk(hipMemcpy(r, d_r, sizeof r, hipMemcpyDeviceToHost));
        snprintf(tag, sizeof tag, "%s device_resid", prec);
        report(tag, it, r);
        // stair preconditioner formed on the device, read back and printed: the test runs the oracle on exactly
        // this Pinv (the one the device solve used), so the iteration counts can be compared for equality
        gbdpcg_handle_t h = gbdpcg_detail::handle();
        if constexpr (sizeof(T) == 4) {
            GBDPCG_CHECK(gbdpcg_form_pinv_f32(h, n, N, 1, (const float *)d_S, (float *)d_P, GBDPCG_PINV_STAIR, nullptr), "form_pinv");
        } else {
            GBDPCG_CHECK(gbdpcg_form_pinv_f64(h, n, N, 1, (const double *)d_S, (double *)d_P, GBDPCG_PINV_STAIR, nullptr), "form_pinv");
        }
        gpuErrchk(hipMemset(d_l, 0, sizeof h_gamma));
        it = solvePCG<T>(n, N, d_S, d_P, d_g, d_l, d_r, d_p, d_v, d_e, &cfg);
        gpuErrchk(hipMemcpy(lam, d_l, sizeof lam, hipMemcpyDeviceToHost));
        snprintf(tag, sizeof tag, "%s device_stair", prec);
        report(tag, it, lam);
        T Pback[36];
        gpuErrchk(hipMemcpy(Pback, d_P, sizeof Pback, hipMemcpyDeviceToHost));
        printf("%s device_stair_pinv=", prec);
        for (int i = 0; i < 36; ++i) printf("%.17g ", (double)Pback[i]);
        printf("\n");
        for (T *q : {d_S, d_P, d_g, d_l, d_r, d_p, d_v, d_e}) gpuErrchk(hipFree(q));
    }
    {   // CSR overload: dense-ish CSR of the same matrix (zeros of the pattern included)
        std::vector<uint32_t> row_ptr(1, 0), col;
        std::vector<T> val;
        for (uint32_t row = 0; row < n * N; ++row) {
            const uint32_t k = row / n, r = row % n;
            for (uint32_t b = 0; b < 3; ++b) {
                if ((k == 0 && b == 0) || (k == N - 1 && b == 2)) continue;
                for (uint32_t c = 0; c < n; ++c) {
                    col.push_back((k + b - 1) * n + c);
                    val.push_back(h_S[k * 12 + b * 4 + c * n + r]);
                }
            }
            row_ptr.push_back((uint32_t)col.size());
        }
        csr_t<T> S{row_ptr.data(), col.data(), val.data(), n * N, n * N, (uint32_t)val.size()};
        T lam[6] = {0};
        pcg_config<T> cfg;
        uint32_t it = solvePCG<T>(&S, (csr_t<T> *)nullptr, h_gamma, lam, n, N, &cfg);
        snprintf(tag, sizeof tag, "%s csr_ident", prec);
        report(tag, it, lam);
    }
}

int main()
{
    run<double>("f64");
    run<float>("f32");
    return 0;
}
