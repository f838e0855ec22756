/*
 * gbdpcg.h -- C ABI of libgbdpcg.so: MI355X-native (gfx950) block-tridiagonal PCG.
 *
 * This is the drop-in boundary for the solve path of A2R-Lab/GBD-PCG.  Every entry
 * point cites the reference interface it replaces (paths relative to the reference
 * checkout).  Plain pointers and sizes only; no C++ or torch types cross it.  The
 * C++ surface the reference's callers see (solvePCG<T>, pcg_config<T>, ...) is the
 * header-only wrapper include/gbdpcg.hpp, which forwards here.
 *
 * DATA LAYOUT (include/pcg.cuh:104-110, include/utils.cuh:80)
 *   A block-tridiagonal matrix with N block-rows of n x n blocks is one array of
 *   3*n*n*N elements: for knot k the three column-major blocks [L_k | D_k | R_k]
 *   (block-row k, block-columns k-1, k, k+1); element (r,c) of block b lives at
 *   k*3n^2 + b*n^2 + c*n + r.  L_0 and R_{N-1} are present but never read.
 *   Vectors (gamma, lambda, r, p) have n*N elements.
 *   A batch of `batch` independent problems is the problem-major concatenation of
 *   the single-problem layout (matrix stride 3n^2N, vector stride nN).  The
 *   reference has no batch notion: batch = 1 is its case.
 *
 * ERRORS
 *   Every function returns a gbdpcg_status; nothing here prints or exits (the
 *   reference's gpuErrchk / exit(code) convention, include/gpuassert.cuh:5-14, is
 *   reproduced by the C++ wrapper).  Functions taking a stream are asynchronous and
 *   capturable into a hipGraph: they never allocate, free or synchronise.
 *
 * There is no CPU fallback: without a gfx950 device gbdpcg_create fails with
 * GBDPCG_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef GBDPCG_H
#define GBDPCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gbdpcg_status {
    GBDPCG_OK = 0,
    GBDPCG_ERR_INVALID = 1,       /* bad argument (null pointer, n == 0, N == 0, ...) */
    GBDPCG_ERR_HIP = 2,           /* a HIP runtime call failed; see gbdpcg_last_hip_error */
    GBDPCG_ERR_NO_DEVICE = 3,     /* no usable gfx950 device */
    GBDPCG_ERR_UNSUPPORTED = 4,   /* shape outside what this build supports */
    GBDPCG_ERR_TOO_LARGE = 5,     /* checkPcgOccupancy analogue: does not fit on the device */
    GBDPCG_ERR_ALLOC = 6,         /* workspace allocation failed */
    GBDPCG_ERR_NOT_IMPLEMENTED = 12 /* the reference's exit code for its CSR stub (interface.cuh:18-19) */
} gbdpcg_status;

typedef struct gbdpcg_context *gbdpcg_handle_t;
typedef struct gbdpcg_graph *gbdpcg_graph_t;

/* Which execution strategy a solve uses.  AUTO picks by shape. */
typedef enum gbdpcg_path {
    GBDPCG_PATH_AUTO = 0,
    GBDPCG_PATH_FUSED = 1, /* one workgroup per problem, vectors LDS-resident, one launch per solve */
    GBDPCG_PATH_SPLIT = 2, /* many workgroups per problem, two launches per iteration, vectors in L2/HBM */
    GBDPCG_PATH_PERSISTENT = 3, /* one large problem over many CUs in ONE launch: block-rows register-resident for the
                                  whole solve (the reference's layout, pcg.cuh:104-110), two in-kernel all-gathers of
                                  {partial inner product, boundary knots} per iteration instead of 4 grid.sync().  Wants
                                  every workgroup resident at once (ceil(N/K) * batch <= CU count, K <= 4).  The
                                  reference refuses a launch that cannot be co-resident before it starts
                                  (checkPcgOccupancy, pcg.cuh:23-49); here, if another kernel holds compute units for
                                  about two seconds, the workgroups give up after a bounded spin WITHOUT having written
                                  anything, and the one that leaves last solves the problem alone inside the same launch
                                  (streaming kernel: slow, correct).  The caller always gets a solved problem; on a
                                  device shared with long-running kernels GBDPCG_PATH_SPLIT avoids the wait. */
    GBDPCG_PATH_PERSISTENT_1R = 4 /* OPT-IN, never chosen by AUTO: the persistent launch with the single-reduction
                                  (Chronopoulos-Gear) recurrence -- u = Pinv r, w = S u, gamma = r.u and delta = u.w in ONE
                                  all-gather per iteration (the halo knots of u are recomputed, not exchanged),
                                  alpha = gamma / (delta - beta gamma / alpha_old), s = S p by recurrence.  Same iterates and the same exit test as pcg.cuh:154-206 in exact arithmetic,
                                  a different rounding sequence: equal iteration counts and fp64 lambda within 1e-13 of the
                                  default path on the test shapes, but not the reference's recurrence. */
} gbdpcg_path;

/* Preconditioners gbdpcg_form_pinv can build from S. */
typedef enum gbdpcg_pinv_kind {
    GBDPCG_PINV_IDENTITY = 0,
    GBDPCG_PINV_BLOCK_JACOBI = 1, /* diag blocks D_k^-1 */
    GBDPCG_PINV_STAIR = 2         /* symmetric stair: D_k^-1, -D_k^-1 O_k D_{k+-1}^-1 */
} gbdpcg_pinv_kind;

/* ---- lifetime ------------------------------------------------------------------------ */

/* One handle per host thread AND per stream of concurrent work: a handle owns device scratch (status
 * words, the split path's workspace, the per-problem symmetry flags) that every solve and every graph
 * created through it uses, so two solves issued through the same handle must not overlap in time
 * (same stream, or otherwise ordered).  `device` is a HIP ordinal.  Replaces the per-call
 * cudaMalloc/cudaFree of interface.cuh:105-108,140-141.
 * A process may hold handles on several devices (one host thread per handle, or one thread looping over
 * them): every entry point makes the handle's device current for the duration of the call and restores the
 * caller's current device before it returns.  Streams and pointers passed in must belong to the handle's device. */
gbdpcg_status gbdpcg_create(gbdpcg_handle_t *out, int device);
gbdpcg_status gbdpcg_destroy(gbdpcg_handle_t h);

const char *gbdpcg_status_string(gbdpcg_status s);
/* hipError_t (as int) of the last failing HIP call on this handle, and its string. */
int gbdpcg_last_hip_error(gbdpcg_handle_t h);
const char *gbdpcg_last_hip_error_string(gbdpcg_handle_t h);

/* Force a path for subsequent solves on this handle (tests / benchmarks). */
gbdpcg_status gbdpcg_set_path(gbdpcg_handle_t h, gbdpcg_path path);
/* Path AUTO would take for this shape (elem_size 4 or 8).  One refinement is decided per solve, where max_iter is known: a
 * batch reported as GBDPCG_PATH_SPLIT that exceeds ONE persistent launch by a few problems (stateSize 14 ... 36) is cut into
 * persistent launches in a row when those cost less than the split path's 2 max_iter + 4 launches. */
gbdpcg_path gbdpcg_choose_path(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N,
                               uint32_t batch);
/* Compute units a GENERAL-storage problem of this shape is spread over inside the fused path (pcg_cluster.hip: both
 * matrices register-resident for the whole solve, 1-8 workgroups per problem (fp64: 1-4) exchanging inner-product partials and
 * boundary knots twice per iteration): built for stateSize 2 ... 16 and 18 in fp32, 2 ... 16 in fp64, horizons up to eight
 * (fp64: four) times what one workgroup holds (stateSize 14, fp32: 72 < knotPoints <= 576).  0 = the shape has no such form: the
 * single-workgroup resident kernel has it (short horizons of stateSize 2 ... 14), or general storage is streamed every
 * iteration (larger blocks, longer horizons). */
uint32_t gbdpcg_cluster_members(uint32_t elem_size, uint32_t n, uint32_t N);
/* The workgroups of one problem wait for each other inside the kernel (bounded spins), so they should get onto the device
 * together: the launch never has more workgroups than compute units and keeps the members of a problem next to each other
 * in dispatch order, which is enough as long as other kernels on the device finish within about a second.  A problem whose
 * workgroups could not meet within the bound is not lost: nothing of it has been written, and the workgroup of its cluster
 * that leaves last solves it (and the cluster's remaining problems) alone with the streaming kernel, inside the same
 * launch; the other problems of the batch are not affected.  d_max_iter_exit is therefore always 0 or 1.  Setting the
 * environment variable GBDPCG_NO_CLUSTER before the first solve of a process switches the form off (general storage is
 * then streamed every iteration, 3.4x slower at the config-3 shape). */

/* Symmetric storage.  S and Pinv of an MPC Schur system are symmetric block-tridiagonal, i.e. in
 * storage L_{k+1} == R_k^T for every knot (README.md:8; the symmetric-stair preconditioner of
 * gbdpcg_form_pinv_* satisfies it bit for bit whenever S does).  Batched solves can then read only
 * [D_k | R_k] of every block-row (2/3 of the bytes) and form L_{k+1} x_k as R_k^T x_k on the fly: with
 * exactly symmetric storage this multiplies the same numbers as the reference, which always reads L_k
 * (include/utils.cuh:77-83); only the summation order differs.  For stateSize 14, fp32, knotPoints <= 128
 * the halves of BOTH matrices (401 KB) then fit the registers + LDS of one compute unit and are read once
 * per solve instead of once per iteration.
 *   mode 2 (default): the relation is TESTED on the device, bit for bit, per problem, before every
 *           solve (one extra pass over L and R of both matrices: 74 us for 1024 problems of n=14, N=128);
 *           problems that pass run the symmetric kernel, the others the general one.  No host round trip.
 *   mode 1: the caller asserts it; no test.   mode 0: never; always read L.
 * Shapes without a symmetric kernel (and solves without a preconditioner) use the general kernels.
 * gbdpcg_check_symmetric_* exposes the test: d_flags[b] = 1 iff problem b satisfies the relation. */
gbdpcg_status gbdpcg_set_symmetric(gbdpcg_handle_t h, int mode);
gbdpcg_status gbdpcg_check_symmetric_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                         const float *d_M, uint8_t *d_flags, void *stream);
gbdpcg_status gbdpcg_check_symmetric_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                         const double *d_M, uint8_t *d_flags, void *stream);

/* ---- sizing helpers -------------------------------------------------------------------- */

/* pcgSharedMemSize<T> (include/pcg.cuh:13-20): elem_size * max(6n^2 + 10n + 2max(n,N), 9n^2).
 * Kept for callers that print / check it; this library sizes its own LDS. */
size_t gbdpcg_pcg_shared_mem_size(uint32_t elem_size, uint32_t n, uint32_t N);

/* checkPcgOccupancy<T> (include/pcg.cuh:23-49): GBDPCG_OK if a (n, N, batch) solve can run on
 * the handle's device, GBDPCG_ERR_TOO_LARGE otherwise (the reference exits 5/6 instead). */
gbdpcg_status gbdpcg_check_occupancy(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N,
                                     uint32_t batch);

/* Bytes of device workspace the SPLIT path needs for this shape (0 for FUSED).  The handle
 * grows its own workspace (and the verdict bytes of the device symmetry check) on demand outside
 * stream capture; call gbdpcg_reserve first when a solve will be captured into a caller-owned graph.
 * Growth never frees: graphs captured earlier keep the old buffers in their kernel nodes, so a replaced
 * buffer stays allocated until gbdpcg_destroy (sizes at least double, so at most 2x the largest is held).
 * The persistent path keeps one small zero-initialised hand-off workspace per shape it has run (element size, n, N,
 * batch), created on first use outside capture (or by gbdpcg_reserve) and kept until gbdpcg_destroy. */
size_t gbdpcg_workspace_bytes(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N,
                              uint32_t batch);
gbdpcg_status gbdpcg_reserve(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N,
                             uint32_t batch);

/* ---- the hot path ---------------------------------------------------------------------- */

/* y = M x, block-tridiagonal, batched.  Replaces loadbdVec + bdmv (include/utils.cuh:9-85) as a
 * standalone operator; this is the kernel the HBM-roofline target is quoted on.
 * All pointers are device pointers; x and y must not alias. */
gbdpcg_status gbdpcg_spmv_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                              const float *d_M, const float *d_x, float *d_y, void *stream);
gbdpcg_status gbdpcg_spmv_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                              const double *d_M, const double *d_x, double *d_y, void *stream);

/* PCG solve of  Pinv S lambda = Pinv gamma  for `batch` independent problems.
 * Replaces the kernel pcg<T,n,N> (include/pcg.cuh:54-218) and its launch
 * (include/interface.cuh:110-133); same algorithm, same exit test (|eta_new| < tol, absolute),
 * same outputs:
 *   d_lambda  [batch*nN] in: initial guess, out: solution              (pcg.cuh:119,215)
 *   d_r, d_p  [batch*nN] out: final residual / direction, may be NULL  (pcg.cuh:125,175,139,205)
 *   d_iters   [batch]    out: iterations taken                         (pcg.cuh:212)
 *   d_max_iter_exit [batch] out: 1 if the loop ran out, 0 if converged (pcg.cuh:212), may be NULL
 * d_Pinv == NULL means the identity preconditioner.  The reference's scratch arguments
 * d_v_temp / d_eta_new_temp (interface.cuh:101-102) have no counterpart: dot products are
 * reduced on chip.  Asynchronous on `stream`; no host synchronisation. */
gbdpcg_status gbdpcg_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                               const float *d_S, const float *d_Pinv, const float *d_gamma,
                               float *d_lambda, float *d_r, float *d_p, float tol,
                               uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit,
                               void *stream);
gbdpcg_status gbdpcg_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                               const double *d_S, const double *d_Pinv, const double *d_gamma,
                               double *d_lambda, double *d_r, double *d_p, double tol,
                               uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit,
                               void *stream);

/* Single-problem, blocking form of the device-pointer overload solvePCG<T>(state_size,
 * knot_points, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, d_v_temp, d_eta_new_temp, config)
 * (include/interface.cuh:92-144): launches on the null stream, waits, returns the iteration
 * count through *h_iters (what the reference returns at :143) and the max-iter flag the
 * reference computes but never copies back (:107-108,141). */
gbdpcg_status gbdpcg_solve_blocking_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                        const float *d_S, const float *d_Pinv,
                                        const float *d_gamma, float *d_lambda, float *d_r,
                                        float *d_p, float tol, uint32_t max_iter,
                                        uint32_t *h_iters, uint8_t *h_max_iter_exit);
gbdpcg_status gbdpcg_solve_blocking_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                        const double *d_S, const double *d_Pinv,
                                        const double *d_gamma, double *d_lambda, double *d_r,
                                        double *d_p, double tol, uint32_t max_iter,
                                        uint32_t *h_iters, uint8_t *h_max_iter_exit);

/* Host-pointer form of solvePCG<T>(h_S, h_gamma, h_lambda, stateSize, knotPoints, config)
 * (include/interface.cuh:24-89): allocates device buffers, copies S / gamma / lambda in,
 * solves, copies lambda out, frees.  Documented deviations from the reference, whose
 * behaviour here is undefined (it never initialises Pinv, :45-46,57-59, and returns the
 * constant 1, :88): h_Pinv == NULL means the identity preconditioner, and *h_iters receives
 * the real iteration count. */
gbdpcg_status gbdpcg_solve_host_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, const float *h_S,
                                    const float *h_Pinv, const float *h_gamma, float *h_lambda,
                                    float tol, uint32_t max_iter, uint32_t *h_iters,
                                    uint8_t *h_max_iter_exit);
gbdpcg_status gbdpcg_solve_host_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, const double *h_S,
                                    const double *h_Pinv, const double *h_gamma, double *h_lambda,
                                    double tol, uint32_t max_iter, uint32_t *h_iters,
                                    uint8_t *h_max_iter_exit);

/* ---- hipGraph-captured solves ---------------------------------------------------------- */

/* Captures gbdpcg_solve_* with these exact arguments into an executable hipGraph (the
 * "whole loop hipGraph-captured" of the north star).  MPC callers re-solve with the same
 * buffers every control step: build once, gbdpcg_graph_launch per step. */
gbdpcg_status gbdpcg_graph_create_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                            uint32_t batch, const float *d_S, const float *d_Pinv,
                                            const float *d_gamma, float *d_lambda, float *d_r,
                                            float *d_p, float tol, uint32_t max_iter,
                                            uint32_t *d_iters, uint8_t *d_max_iter_exit,
                                            gbdpcg_graph_t *out);
gbdpcg_status gbdpcg_graph_create_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                            uint32_t batch, const double *d_S,
                                            const double *d_Pinv, const double *d_gamma,
                                            double *d_lambda, double *d_r, double *d_p, double tol,
                                            uint32_t max_iter, uint32_t *d_iters,
                                            uint8_t *d_max_iter_exit, gbdpcg_graph_t *out);
gbdpcg_status gbdpcg_graph_launch(gbdpcg_graph_t g, void *stream);
gbdpcg_status gbdpcg_graph_destroy(gbdpcg_graph_t g);

/* ---- either side of the solve ---------------------------------- */

/* Builds Pinv from S on the device (f1): the step the reference's host overload lacks
 * (interface.cuh:33-34,45-46) and MPCGPU does with the block helpers of
 * include/utils.cuh:96-161.  d_Pinv gets the same [L|D|R] layout as d_S. */
gbdpcg_status gbdpcg_form_pinv_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                   const float *d_S, float *d_Pinv, gbdpcg_pinv_kind kind,
                                   void *stream);
gbdpcg_status gbdpcg_form_pinv_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                   const double *d_S, double *d_Pinv, gbdpcg_pinv_kind kind,
                                   void *stream);

/* gbdpcg_form_pinv_* followed by gbdpcg_solve_* on the same stream, as one call: what an SQP step does with a
 * freshly formed S (the host overload of the reference stops short of it, interface.cuh:33-34).  d_Pinv is an
 * OUTPUT here and stays valid afterwards.  In symmetric mode 2, for the shapes the one-launch stair kernel covers
 * (even stateSize <= 16), that kernel already compares L_{k+1} with R_k^T of S pair by pair -- it writes the pair
 * of Pinv as mirror images when they match -- so its per-problem verdict replaces the solve's own test launch
 * (74 us of a 0.30 ms converged solve of 1024 problems, n=14, N=128).  Same results as the two calls.
 * The graph form captures both steps for fixed buffers: replay it after rewriting S and gamma in place. */
gbdpcg_status gbdpcg_form_pinv_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                         const float *d_S, float *d_Pinv, gbdpcg_pinv_kind kind,
                                         const float *d_gamma, float *d_lambda, float *d_r, float *d_p,
                                         float tol, uint32_t max_iter, uint32_t *d_iters,
                                         uint8_t *d_max_iter_exit, void *stream);
gbdpcg_status gbdpcg_form_pinv_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                         const double *d_S, double *d_Pinv, gbdpcg_pinv_kind kind,
                                         const double *d_gamma, double *d_lambda, double *d_r, double *d_p,
                                         double tol, uint32_t max_iter, uint32_t *d_iters,
                                         uint8_t *d_max_iter_exit, void *stream);
gbdpcg_status gbdpcg_graph_create_form_pinv_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                                      uint32_t batch, const float *d_S, float *d_Pinv,
                                                      gbdpcg_pinv_kind kind, const float *d_gamma,
                                                      float *d_lambda, float *d_r, float *d_p, float tol,
                                                      uint32_t max_iter, uint32_t *d_iters,
                                                      uint8_t *d_max_iter_exit, gbdpcg_graph_t *out);
gbdpcg_status gbdpcg_graph_create_form_pinv_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N,
                                                      uint32_t batch, const double *d_S, double *d_Pinv,
                                                      gbdpcg_pinv_kind kind, const double *d_gamma,
                                                      double *d_lambda, double *d_r, double *d_p,
                                                      double tol, uint32_t max_iter, uint32_t *d_iters,
                                                      uint8_t *d_max_iter_exit, gbdpcg_graph_t *out);

/* The MPCGPU steps either side of the solve (f4): forming S and gamma from the KKT blocks of a batch of linearised
 * MPC problems, and recovering the primal step from lambda.  The reference tree has no code for them (README.md:2-11
 * states the system that comes out, README.md:66-77 cites the paper), so the convention is fixed here:
 *
 *     minimise    sum_k  1/2 x_k' Q_k x_k + q_k' x_k  +  sum_{k<N-1} 1/2 u_k' R_k u_k + r_k' u_k
 *     subject to  x_0 = c_0,    x_{k+1} - A_k x_k - B_k u_k = c_{k+1}                       (k = 0 .. knotPoints-1)
 *
 * i.e. 1/2 z'Gz + g'z subject to Cz = c, z = (x_0, u_0, x_1, ..., x_{N-1}); Gz + g + C'lambda = 0 gives
 *     S lambda = gamma,  S = C G^-1 C',  gamma = -(c + C G^-1 g),      z = -G^-1 (g + C' lambda)
 * with D_0 = Q_0^-1, D_k = A_j Q_j^-1 A_j' + B_j R_j^-1 B_j' + Q_k^-1 (j = k-1), L_k = -A_j Q_j^-1, R_k = L_{k+1}'.
 * Packed device arrays, one problem after the other, every block column-major (nx = stateSize, nu = controlSize):
 *     d_G    [Q_0 R_0 Q_1 R_1 ... Q_{N-1}]   (nx^2+nu^2) N - nu^2     cost Hessians, symmetric positive definite
 *     d_C    [A_0 B_0 A_1 B_1 ... B_{N-2}]   (nx^2+nx nu)(N-1)        dynamics Jacobians (A: nx x nx, B: nx x nu)
 *     d_g    [q_0 r_0 q_1 r_1 ... q_{N-1}]   (nx+nu) N - nu           cost gradients; d_z has this layout too
 *     d_c    [c_0 ... c_{N-1}]               nx N                     constraint residuals
 *     d_S, d_gamma                            3 nx^2 N, nx N           what gbdpcg_solve_* takes (n = nx)
 *     d_Ginv                                  as d_G                   every block inverted (may be NULL in form_schur)
 * The S written is exactly symmetric in storage (L_{k+1} == R_k' bit for bit), so the default symmetric mode of the solve
 * takes its resident kernels.  Shapes whose per-row working set exceeds one compute unit's LDS (7 nx^2 + 3 nu^2 elements
 * > 160 KB) give GBDPCG_ERR_UNSUPPORTED. */
gbdpcg_status gbdpcg_form_schur_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                    const float *d_G, const float *d_C, const float *d_g, const float *d_c,
                                    float *d_S, float *d_gamma, float *d_Ginv, void *stream);
gbdpcg_status gbdpcg_form_schur_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                    const double *d_G, const double *d_C, const double *d_g, const double *d_c,
                                    double *d_S, double *d_gamma, double *d_Ginv, void *stream);
gbdpcg_status gbdpcg_recover_primal_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                        const float *d_Ginv, const float *d_C, const float *d_g,
                                        const float *d_lambda, float *d_z, void *stream);
gbdpcg_status gbdpcg_recover_primal_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                        const double *d_Ginv, const double *d_C, const double *d_g,
                                        const double *d_lambda, double *d_z, void *stream);

/* One inner step of the SQP loop as ONE call (or one hipGraph for fixed buffers): gbdpcg_form_schur_* writes S, gamma and
 * G^-1, gbdpcg_form_pinv_solve_* forms Phi^-1 from that S (the S of form_schur is symmetric in storage, so in the default
 * symmetric mode the solve runs its resident symmetric kernels without a test launch of its own) and iterates from the
 * d_lambda it finds (warm start), gbdpcg_recover_primal_* writes the primal step z.  Same results as the three calls.
 * Every buffer is the caller's (layouts above); d_r, d_p may be NULL as in gbdpcg_solve_*.  Replay the graph after
 * rewriting G, C, g, c (and lambda, if no warm start is wanted) in place. */
gbdpcg_status gbdpcg_kkt_step_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const float *d_G,
                                  const float *d_C, const float *d_g, const float *d_c, float *d_S, float *d_gamma,
                                  float *d_Ginv, float *d_Pinv, gbdpcg_pinv_kind kind, float *d_lambda, float *d_r, float *d_p,
                                  float tol, uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit, float *d_z,
                                  void *stream);
gbdpcg_status gbdpcg_kkt_step_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const double *d_G,
                                  const double *d_C, const double *d_g, const double *d_c, double *d_S, double *d_gamma,
                                  double *d_Ginv, double *d_Pinv, gbdpcg_pinv_kind kind, double *d_lambda, double *d_r,
                                  double *d_p, double tol, uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit,
                                  double *d_z, void *stream);
gbdpcg_status gbdpcg_graph_create_kkt_step_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                               const float *d_G, const float *d_C, const float *d_g, const float *d_c,
                                               float *d_S, float *d_gamma, float *d_Ginv, float *d_Pinv, gbdpcg_pinv_kind kind,
                                               float *d_lambda, float *d_r, float *d_p, float tol, uint32_t max_iter,
                                               uint32_t *d_iters, uint8_t *d_max_iter_exit, float *d_z, gbdpcg_graph_t *out);
gbdpcg_status gbdpcg_graph_create_kkt_step_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                               const double *d_G, const double *d_C, const double *d_g, const double *d_c,
                                               double *d_S, double *d_gamma, double *d_Ginv, double *d_Pinv,
                                               gbdpcg_pinv_kind kind, double *d_lambda, double *d_r, double *d_p, double tol,
                                               uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit, double *d_z,
                                               gbdpcg_graph_t *out);

/* CSR ingestion (f3): repacks a host CSR matrix (csr_t<T>, include/types.cuh:7-15) whose
 * sparsity lies inside the block-tridiagonal pattern into the [L|D|R] layout (host arrays).
 * Entries outside the pattern give GBDPCG_ERR_INVALID.  Implements what the stub overload
 * at include/interface.cuh:8-20 announces. */
gbdpcg_status gbdpcg_csr_to_bt_f32(uint32_t n, uint32_t N, const uint32_t *row_ptr,
                                   const uint32_t *col_ind, const float *val, float *h_M);
gbdpcg_status gbdpcg_csr_to_bt_f64(uint32_t n, uint32_t N, const uint32_t *row_ptr,
                                   const uint32_t *col_ind, const double *val, double *h_M);

/* Library / build identification ("gbdpcg <version> gfx950"). */
const char *gbdpcg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GBDPCG_H */
