// pcg_stream.hpp -- one workgroup solves one problem by STREAMING S and Pinv from memory every iteration.
//
// The body of pcg_fused_kernel (pcg_fused.hip), as a device function, so that the kernels whose workgroups wait for each
// other (pcg_cluster.hip, pcg_persist.hip) can fall back on it INSIDE their own launch: when the workgroups of a problem
// could not meet (somebody else's kernel holds compute units), the one that finishes last -- an agent-scope counter
// tells it; every other workgroup of the problem has left by then, and none of them has written anything -- solves the
// problem alone from the untouched inputs (stream_rescue below).  No caller ever sees an unsolved problem, the solve stays
// ONE kernel node of a hipGraph, and a healthy launch pays nothing.  (The reference refuses a launch that cannot be
// co-resident before it starts instead: checkPcgOccupancy, /root/reference/include/pcg.cuh:23-49.)
//
// Iteration restated from pcg.cuh:118-208 (see oracle/pcg_oracle_impl.inc for the sequential form).
#pragma once

#include "bt_device.hpp"
#include "bt_sym.hpp"
#include "internal.hpp"

namespace gbdpcg {

// The matrices are re-read every iteration and largely served by the Infinity Cache: default policy.
#ifndef GBDPCG_PCG_NT
#define GBDPCG_PCG_NT 0
#endif
constexpr bool kPcgNT = GBDPCG_PCG_NT != 0;

// Vector carve (elements of T), every array 16-byte aligned:
//   xa  (N+2)n   padded SpMV input: lambda in the prologue, then p      (pads stay zero)
//   xb  (N+2)n   padded SpMV input: r
//   yc  N n      SpMV output: S lambda, then upsilon = S p, then r~ = Pinv r
//   lam N n      lambda
//   red 2*WAVES  per-wave partials of the two inner products
//   zc  N n      (symmetric streaming only) the transposed products R_{k-1}^T x_{k-1}, added into yc
template <typename T> struct FusedCarve {
    uint32_t xa, xb, yc, lam, red, zc, total;
    __host__ __device__ FusedCarve(uint32_t n, uint32_t N, uint32_t waves, bool sym = false) {
        const uint32_t padded = align16<T>((N + 2) * n), plain = align16<T>(N * n);
        xa = 0;
        xb = xa + padded;
        yc = xb + padded;
        lam = yc + plain;
        red = lam + plain;
        zc = red + align16<T>(2 * waves);
        total = zc + (sym ? plain : 0u);
    }
};

// y = M * X (X padded in LDS) for the block-rows of this wave, out of an already primed stream;
// returns this LANE's partial of dot(y, D) where D is a padded LDS vector (D + n = first element).
template <typename T, int NCT, int V>
__device__ __forceinline__ T wg_spmv_dot(RowStream<T, NCT, V, kPcgNT> &rs, const T *X, T *Y, const T *D,
                                         const LaneMap<NCT, V> &m, const StreamCtx<T, NCT, V> &cx, uint32_t n,
                                         uint32_t N, uint32_t lane)
{
    T part = T(0);
    rs.run(X, 0u, N, m, cx, lane, [&](uint32_t k, const T(&acc)[V]) __attribute__((always_inline)) {
        if (m.active && m.g == 0) {
            const uint32_t row = k * n + m.rp * V;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                Y[row + v] = acc[v];
                part = fma_t(acc[v], D[n + row + v], part);
            }
        }
    });
    return part;
}

// Workgroup-wide sum of per-lane partials; every thread returns the same bits.
// Ends with a barrier-protected read, so Y written before the call is visible after it.
template <typename T, int WAVES>
__device__ __forceinline__ T wg_sum(T part, T *red, uint32_t lane, uint32_t wave)
{
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    T tot = red[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) tot += red[w];
    return tot;
}

// Everything a workgroup of WAVES waves needs to stream problems: lane maps and the two register rings, made once.
template <typename T, int NCT, int V, int WAVES, bool SYM> struct StreamSolver {
    static constexpr uint32_t THREADS = WAVES * 64;
    uint32_t n, tid, lane, wave;
    LaneMap<NCT, V> m;
    StreamCtx<T, NCT, V> cx;
    RowStream<T, NCT, V, kPcgNT> rs;
    SymCtx<T, NCT> scx;
    SymStream<T, NCT, kPcgNT> ss;

    __device__ __forceinline__ StreamSolver(uint32_t n_, uint32_t tid_)
        : n(n_), tid(tid_), lane(tid_ & 63u), wave(__builtin_amdgcn_readfirstlane(tid_ >> 6)), m(n_, tid_ & 63u), cx(m, tid_ & 63u),
          scx(tid_ & 63u) {}

    // Solve problem `prob` of `a`.  xa, xb, yc, lam (and zc when SYM): this workgroup's vectors per FusedCarve -- LDS, or
    // device memory touched by this workgroup only (__syncthreads() orders a workgroup's global accesses on its CU);
    // red0: 2 * WAVES partials in LDS.  All threads of the workgroup call it together; it ends on a barrier.
    __device__ __forceinline__ void solve(const PcgArgs<T> &a, uint32_t prob, T *xa, T *xb, T *yc, T *lam, T *red0, T *zc)
    {
        const uint32_t N = a.N, len = n * N;
        T *red1 = red0 + WAVES;
        const size_t mstride = (size_t)3 * n * n * N;
        const T *S = a.S + prob * mstride;
        const T *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;
        const T *gamma = a.gamma + (size_t)prob * len;
        T *lambda = a.lambda + (size_t)prob * len;

        // first matrix loads go out before anything else touches memory
        if constexpr (SYM) ss.prime(S, wave, N, WAVES, scx); else rs.prime(S, wave, N, WAVES, cx, n);
        for (uint32_t i = tid; i < n; i += THREADS) {
            xa[i] = T(0); xa[n + len + i] = T(0);
            xb[i] = T(0); xb[n + len + i] = T(0);
            if constexpr (SYM) zc[i] = T(0);  // row 0 has no block-row above it
        }
        for (uint32_t i = tid; i < len; i += THREADS) {
            const T l = lambda[i];
            xa[n + i] = l;
            lam[i] = l;
        }
        __syncthreads();

        // The solve is a sequence of matrix phases with ONE streaming call site:
        //   phase 0        : yc = S lambda            -> r = gamma - yc                (pcg.cuh:118-126)
        //   phase 1        : yc = Pinv r, eta = r.yc  -> p = yc                        (pcg.cuh:130-149)
        //   phase 2+2i     : yc = S p,    v = p.yc    -> alpha; lambda += alpha p; r -= alpha yc   (:156-176)
        //   phase 3+2i     : yc = Pinv r, eta' = r.yc -> exit test; beta; p = yc + beta p          (:180-206)
        // Each phase primes its own ring at its top.  Priming the NEXT phase's ring before the
        // reduction / update barriers (-DGBDPCG_EARLY_PRIME) was measured 2.6 % slower on config 3
        // (A/B on one device, profiles/r01_ab_prime.txt): two workgroups per CU already cover each
        // other's barrier gaps and the early loads only lengthen the reduction's critical path.
        uint32_t iter = 0;
        bool max_iter_exit = true;
        T eta = T(0);
        for (uint32_t phase = 0;; ++phase) {
            const bool precond = phase & 1u;
            const T *X = precond ? xb : xa;
#ifndef GBDPCG_EARLY_PRIME
            if (phase > 0 && !(precond && !P)) {
                if constexpr (SYM) ss.prime(precond ? P : S, wave, N, WAVES, scx);
                else rs.prime(precond ? P : S, wave, N, WAVES, cx, n);
            }
#endif
            T part = T(0);
            if (precond && !P) {  // identity preconditioner: r~ = r (the primed S units stay in flight)
                for (uint32_t i = tid; i < len; i += THREADS) {
                    const T rv = xb[n + i];
                    yc[i] = rv;
                    part = fma_t(rv, rv, part);
                }
            } else if constexpr (SYM) {
                ss.run(X + n, N, scx,
                       [&](uint32_t k, T a0, T a1) __attribute__((always_inline)) {
                           if (scx.g == 0 && scx.act) {
                               using P2 = typename VecOf<T, 2>::type;
                               P2 v2; v2.x = a0; v2.y = a1;
                               *reinterpret_cast<P2 *>(yc + k * n + scx.rp * 2) = v2;  // n even: 2-element aligned
                           }
                       },
                       [&](uint32_t k, uint32_t c, T t) __attribute__((always_inline)) {
                           if (scx.rp == 0) zc[(k + 1) * n + c - n] = t;
                       });
                __syncthreads();
                // y = (D x_k + R x_{k+1}) + R_{k-1}^T x_{k-1}; the inner product needs the complete y
                for (uint32_t i = tid; i < len; i += THREADS) {
                    const T yv = yc[i] + zc[i];
                    yc[i] = yv;
                    part = fma_t(yv, X[n + i], part);
                }
            } else {
                part = wg_spmv_dot<T, NCT, V>(rs, X, yc, X, m, cx, n, N, lane);
#ifdef GBDPCG_EARLY_PRIME
                // next phase streams the other matrix (or S again under the identity preconditioner)
                rs.prime((precond || !P) ? S : P, wave, N, WAVES, cx, n);
#endif
            }
            if (phase == 0) {
                __syncthreads();
                for (uint32_t i = tid; i < len; i += THREADS) xb[n + i] = gamma[i] - yc[i];
                __syncthreads();
                continue;
            }
            const T tot = wg_sum<T, WAVES>(part, precond ? red1 : red0, lane, wave);
            if (!precond) {
                const T alpha = eta / tot;
                for (uint32_t i = tid; i < len; i += THREADS) {
                    lam[i] = fma_t(alpha, xa[n + i], lam[i]);
                    xb[n + i] = fma_t(-alpha, yc[i], xb[n + i]);
                }
                __syncthreads();
                continue;
            }
            if (phase == 1) {
                eta = tot;
                for (uint32_t i = tid; i < len; i += THREADS) xa[n + i] = yc[i];
                __syncthreads();
                if (a.max_iter == 0) break;
                continue;
            }
            if (fabs(tot) < a.tol) {  // pcg.cuh:195 (absolute test on r.Pinv r)
                ++iter;
                max_iter_exit = false;
                break;
            }
            const T beta = tot / eta;
            eta = tot;
            for (uint32_t i = tid; i < len; i += THREADS) xa[n + i] = fma_t(beta, xa[n + i], yc[i]);
            __syncthreads();
            if (++iter >= a.max_iter) break;
        }

        // ---- outputs   (pcg.cuh:212,215; d_r / d_p as left by :175,:205)
        __syncthreads();
        for (uint32_t i = tid; i < len; i += THREADS) {
            lambda[i] = lam[i];
            if (a.r) a.r[(size_t)prob * len + i] = xb[n + i];
            if (a.p) a.p[(size_t)prob * len + i] = xa[n + i];
        }
        if (tid == 0) {
            a.iters[prob] = iter;
            if (a.max_iter_exit) a.max_iter_exit[prob] = max_iter_exit ? 1 : 0;
        }
        __syncthreads();
    }
};

// ---- in-kernel rescue ---------------------------------------------------------------------------------------------------
// Elements of device memory one rescuing workgroup needs for the vectors of a (n, N) problem (xa, xb, yc, lam of FusedCarve).
template <typename T> __host__ __device__ inline size_t rescue_vec_elems(uint32_t n, uint32_t N)
{
    return FusedCarve<T>(n, N, 1).red;
}

// The calling workgroup (WAVES waves, all threads) solves problem `prob` alone, from the caller's untouched lambda: runtime
// block size, one element per lane and load (any alignment), vectors in `vec` (rescue_vec_elems elements of device memory
// that nobody else touches), partials in `red` (2 * WAVES elements of LDS).  Slow -- one CU's share of the fabric -- and
// only ever reached when a launch could not get its workgroups onto the device together.
template <typename T, int WAVES>
__device__ __forceinline__ void stream_rescue(const PcgArgs<T> &a, uint32_t prob, T *vec, T *red)
{
    const FusedCarve<T> cv(a.n, a.N, 1);
    StreamSolver<T, 0, 1, WAVES, false> sv(a.n, threadIdx.x);
    sv.solve(a, prob, vec + cv.xa, vec + cv.xb, vec + cv.yc, vec + cv.lam, red, nullptr);
}

}  // namespace gbdpcg
