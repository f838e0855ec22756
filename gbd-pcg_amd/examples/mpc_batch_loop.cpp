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

static constexpr uint32_t n = 14;  // stateSize of the reference's iiwa example

static double urand(uint64_t &s)  // splitmix64 -> (-1, 1)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 4503599627370496.0 - 1.0;
}

// S = G W G^T, G = I on the block diagonal and -A_k below it, W_k = I + M_k M_k^T; layout [L_k | D_k | R_k], column-major blocks
static void make_problem(uint32_t N, uint64_t seed, float *S, float *gamma)
{
    std::vector<double> A((size_t)N * n * n), W((size_t)N * n * n), tmp(n * n), D(n * n), O(n * n);
    for (uint32_t k = 0; k < N; ++k) {
        double M[n * n];
        for (double &m : M) m = urand(seed) / std::sqrt((double)n);
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                double w = r == c ? 1.0 : 0.0;
                for (uint32_t q = 0; q < n; ++q) w += M[r * n + q] * M[c * n + q];
                W[(size_t)k * n * n + r * n + c] = w;
                A[(size_t)k * n * n + r * n + c] = 0.35 * urand(seed);  // contraction: keeps S well conditioned
            }
    }
    auto at = [&](const std::vector<double> &X, uint32_t k, uint32_t r, uint32_t c) { return X[(size_t)k * n * n + r * n + c]; };
    for (uint32_t k = 0; k < N; ++k) {
        // D_k = W_k + A_k W_{k-1} A_k^T ;  O_k = S_{k,k+1} = -W_k A_{k+1}^T
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                double d = at(W, k, r, c);
                if (k > 0)
                    for (uint32_t p = 0; p < n; ++p)
                        for (uint32_t q = 0; q < n; ++q) d += at(A, k, r, p) * at(W, k - 1, p, q) * at(A, k, c, q);
                D[r * n + c] = d;
                double o = 0;
                if (k + 1 < N)
                    for (uint32_t q = 0; q < n; ++q) o -= at(W, k, r, q) * at(A, k + 1, c, q);
                O[r * n + c] = o;
            }
        float *blk = S + (size_t)k * 3 * n * n;
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                blk[n * n + c * n + r] = (float)(0.5 * (D[r * n + c] + D[c * n + r]));  // D_k, symmetrised
                blk[2 * n * n + c * n + r] = (float)O[r * n + c];                        // R_k
                if (k + 1 < N) blk[3 * n * n + r * n + c] = (float)O[r * n + c];         // L_{k+1} = R_k^T, bit for bit
            }
        if (k == 0)
            for (uint32_t i = 0; i < n * n; ++i) blk[i] = 0.f;  // L_0: never read
        if (k == N - 1)
            for (uint32_t i = 0; i < n * n; ++i) blk[2 * n * n + i] = 0.f;  // R_{N-1}: never read
        for (uint32_t r = 0; r < n; ++r) gamma[(size_t)k * n + r] = (float)urand(seed);
    }
}

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
