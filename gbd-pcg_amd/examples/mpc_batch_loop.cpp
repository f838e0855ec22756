// Batched "control loop" against the C ABI (include/gbdpcg.h): what an MPC pipeline that solves many
// Schur systems per control step does with this library instead of one solvePCG<T> call per problem
// (/root/reference/include/interface.cuh:92-144 allocates, launches, copies back and frees per call):
//
//   once      : buffers, a handle, one executable graph of { Phi^-1 = symmetric stair from S ; solve } for those buffers
//               (gbdpcg_graph_create_form_pinv_solve_f32: the stair kernel's symmetry verdicts feed the solve, which
//               then needs no test launch of its own)
//   per step  : (the caller rewrites S and gamma in place), replay the graph, read lambda / iteration counts when
//               they are needed
//
// Builds synthetic symmetric positive definite block-tridiagonal systems on the host (S = G W G^T with a
// block-bidiagonal G, the structure of an MPC Schur complement), runs a few steps and prints per-step times
// and the true residual of one problem.   usage: mpc_batch_loop [batch=1024] [knotPoints=128] [steps=5]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gbdpcg.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)
#define GK(x)                                                                                        \
    do {                                                                                             \
        gbdpcg_status s_ = (x);                                                                      \
        if (s_ != GBDPCG_OK) {                                                                       \
            fprintf(stderr, "gbdpcg error %s at %s:%d\n", gbdpcg_status_string(s_), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

#include "synth_problem.hpp"

int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024, N = argc > 2 ? (uint32_t)atoi(argv[2]) : 128;
    const int steps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t msz = (size_t)3 * n * n * N, vsz = (size_t)n * N;

    std::vector<float> hS(msz * batch), hg(vsz * batch);
    const uint32_t distinct = batch < 16 ? batch : 16;  // a few distinct systems, repeated: host generation is O(n^4 N)
    for (uint32_t b = 0; b < distinct; ++b) make_problem(N, 1234 + b, hS.data() + b * msz, hg.data() + b * vsz);
    for (uint32_t b = distinct; b < batch; ++b) {
        std::copy(hS.begin() + (b % distinct) * msz, hS.begin() + (b % distinct + 1) * msz, hS.begin() + b * msz);
        std::copy(hg.begin() + (b % distinct) * vsz, hg.begin() + (b % distinct + 1) * vsz, hg.begin() + b * vsz);
    }

    float *dS, *dP, *dg, *dl;
    uint32_t *d_iters;
    uint8_t *d_flags;
    CK(hipMalloc((void **)&dS, msz * batch * 4));
    CK(hipMalloc((void **)&dP, msz * batch * 4));
    CK(hipMalloc((void **)&dg, vsz * batch * 4));
    CK(hipMalloc((void **)&dl, vsz * batch * 4));
    CK(hipMalloc((void **)&d_iters, batch * 4));
    CK(hipMalloc((void **)&d_flags, batch));
    CK(hipMemcpy(dS, hS.data(), msz * batch * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), vsz * batch * 4, hipMemcpyHostToDevice));

    gbdpcg_handle_t h;
    GK(gbdpcg_create(&h, 0));
    hipStream_t stream;
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    gbdpcg_graph_t graph;  // default symmetric mode: problems whose S has L_{k+1} == R_k^T take the CU-resident path
    GK(gbdpcg_graph_create_form_pinv_solve_f32(h, n, N, batch, dS, dP, GBDPCG_PINV_STAIR, dg, dl, nullptr, nullptr, 1e-6f, 50,
                                               d_iters, d_flags, &graph));

    std::vector<uint32_t> iters(batch);
    for (int step = 0; step < steps; ++step) {
        // (an MPC pipeline would rewrite dS / dg here from the new linearisation)
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipMemsetAsync(dl, 0, vsz * batch * 4, stream));  // cold start; a warm start keeps the previous lambda
        GK(gbdpcg_graph_launch(graph, stream));
        CK(hipStreamSynchronize(stream));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        CK(hipMemcpy(iters.data(), d_iters, batch * 4, hipMemcpyDeviceToHost));
        uint32_t lo = iters[0], hi = iters[0];
        for (uint32_t v : iters) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        printf("step %d: %u problems (n=%u, N=%u): Pinv + solve %.1f us, %.2f us per problem, iterations %u..%u\n", step, batch, n,
               N, us, us / batch, lo, hi);
    }

    // true residual of problem 0 through the library's own block-tridiagonal product
    float *dy;
    CK(hipMalloc((void **)&dy, vsz * 4));
    GK(gbdpcg_spmv_f32(h, n, N, 1, dS, dl, dy, stream));
    CK(hipStreamSynchronize(stream));
    std::vector<float> y(vsz);
    CK(hipMemcpy(y.data(), dy, vsz * 4, hipMemcpyDeviceToHost));
    double rr = 0, gg = 0;
    for (size_t i = 0; i < vsz; ++i) {
        rr += (double)(hg[i] - y[i]) * (hg[i] - y[i]);
        gg += (double)hg[i] * hg[i];
    }
    printf("problem 0: ||gamma - S lambda|| / ||gamma|| = %.3e\n", std::sqrt(rr / gg));

    GK(gbdpcg_graph_destroy(graph));
    GK(gbdpcg_destroy(h));
    return std::sqrt(rr / gg) < 1e-3 ? 0 : 2;
}
