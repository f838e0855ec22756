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
    {   // the steps either side of the solve through the template layer: a 2-state, 1-control, 3-knot problem, printed for the
        // test to hold against a dense solve of the whole KKT system ("kkt_z": z = (x_0, u_0, x_1, u_1, x_2); "kkt_step": lambda)
        const uint32_t nx = 2, nu = 1, Nk = 3;
        const T hG[] = {2, 0.5, 0.5, 1, 3,   1.5, 0.2, 0.2, 2, 1,   1, 0, 0, 4};           // Q_0 R_0 Q_1 R_1 Q_2 (column-major)
        const T hC[] = {1, 0.1, -0.2, 0.9, 0.5, 1,   0.8, 0, 0.3, 1.1, 0, 0.7};             // A_0 B_0 A_1 B_1
        const T hg[] = {1, -1, 0.5,   0.3, 0.2, -0.4,   -0.6, 0.9};                         // q_0 r_0 q_1 r_1 q_2
        const T hc[] = {0.5, -0.25,   0.1, 0.2,   -0.3, 0.05};                              // c_0 c_1 c_2
        T *dG, *dC, *dg, *dc, *dS, *dgam, *dGi, *dP, *dl, *dz;
        uint32_t *d_it;
        uint8_t *d_fl;
        gpuErrchk(hipMalloc((void **)&dG, sizeof hG));
        gpuErrchk(hipMalloc((void **)&dC, sizeof hC));
        gpuErrchk(hipMalloc((void **)&dg, sizeof hg));
        gpuErrchk(hipMalloc((void **)&dc, sizeof hc));
        gpuErrchk(hipMalloc((void **)&dS, 3 * nx * nx * Nk * sizeof(T)));
        gpuErrchk(hipMalloc((void **)&dgam, nx * Nk * sizeof(T)));
        gpuErrchk(hipMalloc((void **)&dGi, sizeof hG));
        gpuErrchk(hipMalloc((void **)&dP, 3 * nx * nx * Nk * sizeof(T)));
        gpuErrchk(hipMalloc((void **)&dl, nx * Nk * sizeof(T)));
        gpuErrchk(hipMalloc((void **)&dz, sizeof hg));
        gpuErrchk(hipMalloc((void **)&d_it, sizeof(uint32_t)));
        gpuErrchk(hipMalloc((void **)&d_fl, 1));
        gpuErrchk(hipMemcpy(dG, hG, sizeof hG, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(dg, hg, sizeof hg, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice));
        gpuErrchk(hipMemset(dl, 0, nx * Nk * sizeof(T)));
        pcg_config<T> cfg;
        cfg.pcg_exit_tol = sizeof(T) == 8 ? T(1e-24) : T(1e-12);
        cfg.pcg_max_iter = 50;
        kktStep<T>(nx, nu, Nk, 1, dG, dC, dg, dc, dS, dgam, dGi, dP, dl, dz, d_it, d_fl, &cfg);
        gpuErrchk(hipDeviceSynchronize());
        T lam[6], z[8];
        uint32_t it;
        gpuErrchk(hipMemcpy(lam, dl, sizeof lam, hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(z, dz, sizeof z, hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(&it, d_it, sizeof it, hipMemcpyDeviceToHost));
        snprintf(tag, sizeof tag, "%s kkt_step", prec);
        report(tag, it, lam);
        printf("%s kkt_z=", prec);
        for (int i = 0; i < 8; ++i) printf("%.17g ", (double)z[i]);
        printf("\n");
        // the same through the two separate wrappers, from the lambda just found: the same z, bit for bit
        T *dz2;
        gpuErrchk(hipMalloc((void **)&dz2, sizeof hg));
        formSchur<T>(nx, nu, Nk, 1, dG, dC, dg, dc, dS, dgam, dGi);
        recoverPrimal<T>(nx, nu, Nk, 1, dGi, dC, dg, dl, dz2);
        gpuErrchk(hipDeviceSynchronize());
        T z2[8];
        gpuErrchk(hipMemcpy(z2, dz2, sizeof z2, hipMemcpyDeviceToHost));
        bool same = true;
        for (int i = 0; i < 8; ++i) same = same && z2[i] == z[i];
        printf("%s kkt_wrappers_agree=%d\n", prec, (int)same);
        for (void *p : {(void *)dG, (void *)dC, (void *)dg, (void *)dc, (void *)dS, (void *)dgam, (void *)dGi, (void *)dP, (void *)dl,
                        (void *)dz, (void *)dz2, (void *)d_it, (void *)d_fl})
            gpuErrchk(hipFree(p));
    }
}

int main()
{
    run<double>("f64");
    run<float>("f32");
    return 0;
}
