// gbdpcg.hpp -- the C++ surface callers of A2R-Lab/GBD-PCG see, re-implemented as a thin,
// header-only layer over the C ABI of libgbdpcg.so (include/gbdpcg.h).  Host code only: nothing
// here is device code, it compiles with hipcc or with g++ -D__HIP_PLATFORM_AMD__.
//
// Reference interface -> what this header provides (paths relative to the reference checkout):
//   include/constants.cuh:5-20   STATE_SIZE / KNOT_POINTS macros, pcg_constants::*
//   include/types.cuh:7-15       csr_t<T>
//   include/types.cuh:18-35      pcg_config<T>            (same fields, order and defaults)
//   include/gpuassert.cuh:5-14   gpuAssert / gpuErrchk    (hipError_t; same message, exit(code))
//   include/pcg.cuh:13-20        pcgSharedMemSize<T>
//   include/pcg.cuh:23-49        checkPcgOccupancy<T>     (exit(5) / exit(6) on failure)
//   include/interface.cuh:8-20   solvePCG<T>(csr_t*, csr_t*, ...)        CSR overload
//   include/interface.cuh:24-89  solvePCG<T>(h_S, h_gamma, h_lambda, ...) host overload
//   include/interface.cuh:92-144 solvePCG<T>(n, N, d_S, d_Pinv, ...)      device overload
//   README.md:42-46              pcg_solve<T>, cbtd_t     (the documented names)
//
// Differences a maintainer should know (all documented in INTEGRATION.md):
//   * stateSize / knotPoints are plain runtime arguments; the -DSTATE_SIZE/-DKNOT_POINTS
//     "double declaration" (README.md:63-64) is no longer needed (the macros are kept, unused).
//   * The host overload treats config->empty_pinv != 0 as "identity preconditioner" (the
//     reference leaves Pinv uninitialised there) and returns the real iteration count instead
//     of the constant 1 (interface.cuh:88).  With empty_pinv == 0 it builds the symmetric-stair
//     preconditioner from S on the device.
//   * The CSR overload works (the reference prints NOT IMPLEMENTED and exits 12).
//   * d_v_temp / d_eta_new_temp are accepted and ignored: inner products never leave the chip.
//   * pcg_grid / pcg_block are accepted and ignored, as in the reference (interface.cuh:132).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <type_traits>
#include <vector>

#include "gbdpcg.h"

#ifndef STATE_SIZE
#define STATE_SIZE 3
#endif
#ifndef KNOT_POINTS
#define KNOT_POINTS 3
#endif

namespace pcg_constants {
inline uint32_t DEFAULT_MAX_PCG_ITER = 25;
template <typename T> inline T DEFAULT_EPSILON = static_cast<T>(1e-6);
inline dim3 DEFAULT_GRID(128);
inline dim3 DEFAULT_BLOCK(64);
}  // namespace pcg_constants

template <typename T> struct csr_t {
    uint32_t *row_ptr;
    uint32_t *col_ind;
    T *val;
    uint32_t rows;
    uint32_t cols;
    uint32_t nnz;
};

template <typename T> struct pcg_config {
    T pcg_exit_tol;
    uint32_t pcg_max_iter;
    dim3 pcg_grid;
    dim3 pcg_block;
    int empty_pinv;

    pcg_config(T exit_tol = pcg_constants::DEFAULT_EPSILON<T>, uint32_t max_iter = pcg_constants::DEFAULT_MAX_PCG_ITER,
               dim3 grid = pcg_constants::DEFAULT_GRID, dim3 block = pcg_constants::DEFAULT_BLOCK, int empty_pinv = 1)
        : pcg_exit_tol(exit_tol), pcg_max_iter(max_iter), pcg_grid(grid), pcg_block(block), empty_pinv(empty_pinv) {}
};

// compressed block-tridiagonal storage, as README.md:46 names it: the flat [L|D|R] array
template <typename T> using cbtd_t = T;

inline void gpuAssert(hipError_t code, const char *file, int line, bool abort = true)
{
    if (code != hipSuccess) {
        fprintf(stderr, "GPUassert: %s %s %d\n", hipGetErrorString(code), file, line);
        if (abort) exit(code);
    }
}
#define gpuErrchk(ans) { gpuAssert((ans), __FILE__, __LINE__); }

namespace gbdpcg_detail {

// One library handle per host thread AND per device, created on first use: the handle of the device that is
// current at the call (as the reference launches on whatever device is current, interface.cuh:132), so a
// thread that loops hipSetDevice(g) over the GPUs of a node gets the right handle for the pointers it passes.
inline gbdpcg_handle_t handle()
{
    struct Holder {
        std::vector<gbdpcg_handle_t> by_device;
        ~Holder() { for (gbdpcg_handle_t h : by_device) if (h) gbdpcg_destroy(h); }
    };
    thread_local Holder holder;
    int dev = 0;
    gpuErrchk(hipGetDevice(&dev));
    if ((size_t)dev >= holder.by_device.size()) holder.by_device.resize((size_t)dev + 1, nullptr);
    gbdpcg_handle_t &h = holder.by_device[(size_t)dev];
    if (!h) {
        gbdpcg_status st = gbdpcg_create(&h, dev);
        if (st != GBDPCG_OK) {
            fprintf(stderr, "GBD-PCG: cannot create solver on device %d: %s\n", dev, gbdpcg_status_string(st));
            exit(static_cast<int>(st));
        }
    }
    return h;
}

// C-ABI status -> the reference's error convention (print, exit with the code).
inline void check(gbdpcg_status st, const char *what, const char *file, int line)
{
    if (st == GBDPCG_OK) return;
    gbdpcg_handle_t h = handle();
    if (st == GBDPCG_ERR_HIP) {
        fprintf(stderr, "GPUassert: %s %s %d\n", gbdpcg_last_hip_error_string(h), file, line);
        exit(gbdpcg_last_hip_error(h));
    }
    fprintf(stderr, "GBD-PCG: %s failed: %s (%s:%d)\n", what, gbdpcg_status_string(st), file, line);
    exit(static_cast<int>(st));
}
#define GBDPCG_CHECK(expr, what) gbdpcg_detail::check((expr), (what), __FILE__, __LINE__)

template <typename T> constexpr bool is_f32 = std::is_same<T, float>::value;
template <typename T> constexpr bool is_f64 = std::is_same<T, double>::value;

}  // namespace gbdpcg_detail

template <typename T> size_t pcgSharedMemSize(uint32_t state_size, uint32_t knot_points)
{
    return gbdpcg_pcg_shared_mem_size(sizeof(T), state_size, knot_points);
}

// The reference checks cooperative-launch support and co-residency of knot_points blocks and exits
// 5 / 6 otherwise.  This build needs neither; what can fail is a problem that fits no path.
template <typename T> bool checkPcgOccupancy(void * /*kernel*/, dim3 /*block*/, uint32_t state_size, uint32_t knot_points)
{
    gbdpcg_status st = gbdpcg_check_occupancy(gbdpcg_detail::handle(), sizeof(T), state_size, knot_points, 1);
    if (st != GBDPCG_OK) {
        printf("Too many knot points ([%d]) or too large a state ([%d]) for this device: %s\n", knot_points,
               state_size, gbdpcg_status_string(st));
        exit(6);
    }
    return true;
}

// ---- device-pointer overload (interface.cuh:92-144): returns the iteration count ----------------
template <typename T>
uint32_t solvePCG(const uint32_t state_size, const uint32_t knot_points, T *d_S, T *d_Pinv, T *d_gamma, T *d_lambda,
                  T *d_r, T *d_p, T * /*d_v_temp*/, T * /*d_eta_new_temp*/, struct pcg_config<T> *config)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "solvePCG<T>: T is float or double");
    uint32_t iters = 0;
    uint8_t max_iter_exit = 0;
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_solve_blocking_f32(h, state_size, knot_points, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p,
                                               config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit),
                     "solvePCG");
    } else {
        GBDPCG_CHECK(gbdpcg_solve_blocking_f64(h, state_size, knot_points, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p,
                                               config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit),
                     "solvePCG");
    }
    return iters;
}

// ---- host-pointer overload (interface.cuh:24-89) -------------------------------------------------
template <typename T>
uint32_t solvePCG(T *h_S, T *h_gamma, T *h_lambda, unsigned stateSize, unsigned knotPoints, struct pcg_config<T> *config)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "solvePCG<T>: T is float or double");
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    uint32_t iters = 0;
    uint8_t max_iter_exit = 0;
    if (config->empty_pinv) {
        if constexpr (gbdpcg_detail::is_f32<T>) {
            GBDPCG_CHECK(gbdpcg_solve_host_f32(h, stateSize, knotPoints, h_S, nullptr, h_gamma, h_lambda,
                                               config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit),
                         "solvePCG");
        } else {
            GBDPCG_CHECK(gbdpcg_solve_host_f64(h, stateSize, knotPoints, h_S, nullptr, h_gamma, h_lambda,
                                               config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit),
                         "solvePCG");
        }
        return iters;
    }
    // empty_pinv == 0: build the symmetric-stair preconditioner from S on the device, then solve
    const size_t melems = (size_t)3 * stateSize * stateSize * knotPoints, velems = (size_t)stateSize * knotPoints;
    T *d_S = nullptr, *d_P = nullptr, *d_g = nullptr, *d_l = nullptr;
    gpuErrchk(hipMalloc(reinterpret_cast<void **>(&d_S), melems * sizeof(T)));
    gpuErrchk(hipMalloc(reinterpret_cast<void **>(&d_P), melems * sizeof(T)));
    gpuErrchk(hipMalloc(reinterpret_cast<void **>(&d_g), velems * sizeof(T)));
    gpuErrchk(hipMalloc(reinterpret_cast<void **>(&d_l), velems * sizeof(T)));
    gpuErrchk(hipMemcpy(d_S, h_S, melems * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_g, h_gamma, velems * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_l, h_lambda, velems * sizeof(T), hipMemcpyHostToDevice));
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_form_pinv_f32(h, stateSize, knotPoints, 1, d_S, d_P, GBDPCG_PINV_STAIR, nullptr), "form_pinv");
    } else {
        GBDPCG_CHECK(gbdpcg_form_pinv_f64(h, stateSize, knotPoints, 1, d_S, d_P, GBDPCG_PINV_STAIR, nullptr), "form_pinv");
    }
    iters = solvePCG<T>(stateSize, knotPoints, d_S, d_P, d_g, d_l, static_cast<T *>(nullptr), static_cast<T *>(nullptr),
                        static_cast<T *>(nullptr), static_cast<T *>(nullptr), config);
    gpuErrchk(hipMemcpy(h_lambda, d_l, velems * sizeof(T), hipMemcpyDeviceToHost));
    gpuErrchk(hipFree(d_S));
    gpuErrchk(hipFree(d_P));
    gpuErrchk(hipFree(d_g));
    gpuErrchk(hipFree(d_l));
    return iters;
}

// ---- CSR overload (interface.cuh:8-20; a stub in the reference, implemented here) -----------------
template <typename T>
uint32_t solvePCG(csr_t<T> *h_S, csr_t<T> *h_Pinv, T *h_gamma, T *h_lambda, unsigned stateSize, unsigned knotPoints,
                  struct pcg_config<T> *config)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "solvePCG<T>: T is float or double");
    const size_t melems = (size_t)3 * stateSize * stateSize * knotPoints;
    auto repack = [&](csr_t<T> *m, std::vector<T> &out) {
        out.resize(melems);
        if (m->rows != stateSize * knotPoints || m->cols != stateSize * knotPoints) {
            fprintf(stderr, "GBD-PCG: CSR matrix is %u x %u, expected %u x %u\n", m->rows, m->cols,
                    stateSize * knotPoints, stateSize * knotPoints);
            exit(static_cast<int>(GBDPCG_ERR_INVALID));
        }
        if constexpr (gbdpcg_detail::is_f32<T>) {
            GBDPCG_CHECK(gbdpcg_csr_to_bt_f32(stateSize, knotPoints, m->row_ptr, m->col_ind, m->val, out.data()), "csr_to_bt");
        } else {
            GBDPCG_CHECK(gbdpcg_csr_to_bt_f64(stateSize, knotPoints, m->row_ptr, m->col_ind, m->val, out.data()), "csr_to_bt");
        }
    };
    std::vector<T> S, P;
    repack(h_S, S);
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    uint32_t iters = 0;
    uint8_t max_iter_exit = 0;
    const T *Pptr = nullptr;
    if (h_Pinv) {
        repack(h_Pinv, P);
        Pptr = P.data();
    }
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_solve_host_f32(h, stateSize, knotPoints, S.data(), Pptr, h_gamma, h_lambda,
                                           config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit), "solvePCG");
    } else {
        GBDPCG_CHECK(gbdpcg_solve_host_f64(h, stateSize, knotPoints, S.data(), Pptr, h_gamma, h_lambda,
                                           config->pcg_exit_tol, config->pcg_max_iter, &iters, &max_iter_exit), "solvePCG");
    }
    return iters;
}

// ---- the steps either side of the solve (no counterpart in the reference tree: README.md:2-11 states the system they produce;
// layouts and formulas in gbdpcg.h).  Device pointers, packed KKT blocks of `batch` problems; errors end the process like gpuErrchk.
template <typename T>
void formSchur(uint32_t stateSize, uint32_t controlSize, uint32_t knotPoints, uint32_t batch, const T *d_G, const T *d_C,
               const T *d_g, const T *d_c, T *d_S, T *d_gamma, T *d_Ginv, hipStream_t stream = nullptr)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "formSchur<T>: T is float or double");
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_form_schur_f32(h, stateSize, controlSize, knotPoints, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, stream), "formSchur");
    } else {
        GBDPCG_CHECK(gbdpcg_form_schur_f64(h, stateSize, controlSize, knotPoints, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, stream), "formSchur");
    }
}

template <typename T>
void recoverPrimal(uint32_t stateSize, uint32_t controlSize, uint32_t knotPoints, uint32_t batch, const T *d_Ginv, const T *d_C,
                   const T *d_g, const T *d_lambda, T *d_z, hipStream_t stream = nullptr)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "recoverPrimal<T>: T is float or double");
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_recover_primal_f32(h, stateSize, controlSize, knotPoints, batch, d_Ginv, d_C, d_g, d_lambda, d_z, stream), "recoverPrimal");
    } else {
        GBDPCG_CHECK(gbdpcg_recover_primal_f64(h, stateSize, controlSize, knotPoints, batch, d_Ginv, d_C, d_g, d_lambda, d_z, stream), "recoverPrimal");
    }
}

// KKT blocks -> S, gamma, G^-1 -> symmetric-stair Phi^-1 -> PCG from the d_lambda found in the buffer -> primal step, one call
// (asynchronous on `stream`; d_iters / d_max_iter_exit: one entry per problem, on the device).
template <typename T>
void kktStep(uint32_t stateSize, uint32_t controlSize, uint32_t knotPoints, uint32_t batch, const T *d_G, const T *d_C, const T *d_g,
             const T *d_c, T *d_S, T *d_gamma, T *d_Ginv, T *d_Pinv, T *d_lambda, T *d_z, uint32_t *d_iters,
             uint8_t *d_max_iter_exit, struct pcg_config<T> *config, hipStream_t stream = nullptr)
{
    static_assert(gbdpcg_detail::is_f32<T> || gbdpcg_detail::is_f64<T>, "kktStep<T>: T is float or double");
    gbdpcg_handle_t h = gbdpcg_detail::handle();
    if constexpr (gbdpcg_detail::is_f32<T>) {
        GBDPCG_CHECK(gbdpcg_kkt_step_f32(h, stateSize, controlSize, knotPoints, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, d_Pinv,
                                         GBDPCG_PINV_STAIR, d_lambda, nullptr, nullptr, config->pcg_exit_tol, config->pcg_max_iter,
                                         d_iters, d_max_iter_exit, d_z, stream), "kktStep");
    } else {
        GBDPCG_CHECK(gbdpcg_kkt_step_f64(h, stateSize, controlSize, knotPoints, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, d_Pinv,
                                         GBDPCG_PINV_STAIR, d_lambda, nullptr, nullptr, config->pcg_exit_tol, config->pcg_max_iter,
                                         d_iters, d_max_iter_exit, d_z, stream), "kktStep");
    }
}

// ---- the README's spelling (README.md:42): int pcg_solve<T>(cbtd_t *h_S, ...) ---------------------
template <typename T>
int pcg_solve(cbtd_t<T> *h_S, T *h_gamma, T *h_lambda, unsigned stateSize, unsigned knotPoints,
              pcg_config<T> *config = nullptr)
{
    pcg_config<T> default_config;
    return static_cast<int>(solvePCG<T>(h_S, h_gamma, h_lambda, stateSize, knotPoints, config ? config : &default_config));
}
