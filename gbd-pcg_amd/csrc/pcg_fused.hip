// pcg_fused.hip -- batched PCG, one workgroup per problem, one launch per solve.
//
// Replaces the cooperative kernel pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218).  The
// reference keeps a knot's block-rows of S and Pinv in shared memory, gives every knot a block
// and crosses the grid with 4 grid.sync() per iteration.  On MI355X a grid barrier costs
// 4-26 us (MI355X_MICROARCH.md, barrier-xcd / barrier-cg), so this kernel turns the
// decomposition around: the VECTORS (lambda, r, p, and the S p / Pinv r product) of one
// problem live in one workgroup's LDS for the whole solve, S and Pinv are streamed from HBM
// once per iteration by RowStream, and both inner products are reduced inside the
// workgroup (wave butterfly -> WAVES partials in LDS -> same-order sum in every thread, which
// keeps the convergence branch uniform like pcg.cuh:147,167,191 do).  No cross-CU traffic at
// all; each problem exits on its own iteration count.  Algorithmic HBM bytes per
// problem-iteration: 2 (3N-2) n^2 sizeof(T)  (SURVEY.md section 8d).
//
// Iteration restated from pcg.cuh:118-208 (see oracle/pcg_oracle_impl.inc for the sequential form).
#include <cstdlib>

#include "bt_device.hpp"
#include "bt_sym.hpp"
#include "internal.hpp"

namespace gbdpcg {

// The matrices are re-read every iteration and largely served by the Infinity Cache: default policy.
#ifndef GBDPCG_PCG_NT
#define GBDPCG_PCG_NT 0
#endif
constexpr bool kPcgNT = GBDPCG_PCG_NT != 0;

// LDS carve (elements of T), every array 16-byte aligned:
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

// SYM: both matrices are symmetric block-tridiagonal (L_{k+1} == R_k^T) and are streamed through
// SymStream (bt_sym.hpp): [D_k | R_k] only, 2/3 of the bytes.
// GVEC (rescue launches of problems too large for one workgroup's LDS, see PcgArgs::rescue): the four vectors live in
// device memory (a.rescue_vec, one carve per workgroup) instead of LDS -- written and read by this workgroup only, and
// __syncthreads() orders a workgroup's global accesses on one CU -- and only the wave partials stay in LDS.  One workgroup
// then pulls a problem of any size through one CU: slow (the fabric share of one CU), correct, and one launch.
template <typename T, int NCT, int V, int WAVES, bool SYM, bool GVEC = false>
__global__ __launch_bounds__(WAVES * 64) void pcg_fused_kernel(PcgArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    static_assert(!(GVEC && SYM), "rescue launches stream general storage");

    constexpr uint32_t THREADS = WAVES * 64;
    const uint32_t n = NCT ? (uint32_t)NCT : a.n;
    const uint32_t N = a.N;
    const uint32_t len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps row bases in SGPRs
    const FusedCarve<T> cv(n, N, WAVES, SYM);
    T *vec = GVEC ? reinterpret_cast<T *>(a.rescue_vec) + (size_t)blockIdx.x * cv.red : smem;
    T *xa = vec + cv.xa, *xb = vec + cv.xb, *yc = vec + cv.yc, *lam = vec + cv.lam;
    T *red0 = GVEC ? smem : smem + cv.red, *red1 = red0 + WAVES;
    T *zc = smem + cv.zc;
    const LaneMap<NCT, V> m(n, lane);
    const StreamCtx<T, NCT, V> cx(m, lane);
    RowStream<T, NCT, V, kPcgNT> rs;
    const SymCtx<T, NCT> scx(lane);
    SymStream<T, NCT, kPcgNT> ss;
    const size_t mstride = (size_t)3 * n * n * N;

    unsigned long long rescue_mask = 0ull;   // rescue launches: verdicts of this workgroup's next 64 problems, one round trip
    uint32_t pi = 0;
    for (uint32_t prob = blockIdx.x; prob < a.batch; prob += gridDim.x, ++pi) {
        if (a.rescue) {
            if ((pi & 63u) == 0u) {
                const uint32_t left = (a.batch - prob + gridDim.x - 1) / gridDim.x;
                rescue_mask = pcg_takes_mask(a, prob, gridDim.x, left < 64u ? left : 64u, lane);
            }
            if (!((rescue_mask >> (pi & 63u)) & 1ull)) continue;   // (normally every problem: nothing gave up)
        } else if (!pcg_takes(a, prob)) continue;  // this launch is not the one that owns the problem
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
}

template <typename T> bool fused_has_symmetric(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch)
{
    if (resident_shape<T>(n, N)) return false;         // register-resident kernel reads each matrix once anyway
    if (resident_sym_shape<T>(n, N)) return true;      // symmetric halves fit one CU: pcg_resident_sym.hip
    if (batch < (uint32_t)dev.num_cus) return false;  // small batches: not worth the symmetry check
    bool ok = false;
#define GBDPCG_CASE(NN) \
    if (n == NN) ok = SymGeom<T, NN>::OK && best_v<T, NN>() >= 2;
    GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    return ok && FusedCarve<T>(n, N, 8, true).total * sizeof(T) <= dev.lds_per_wg_max;
}

template <typename T> size_t fused_lds_bytes(uint32_t n, uint32_t N, uint32_t waves)
{
    return (size_t)FusedCarve<T>(n, N, waves).total * sizeof(T);
}

template <typename T> bool fused_fits(const DeviceInfo &dev, uint32_t n, uint32_t N)
{
    return fused_lds_bytes<T>(n, N, 16) <= dev.lds_per_wg_max;
}

// Workgroups of a rescue launch: it normally owns nothing (one round trip per workgroup to find that out) and otherwise a
// handful of problems, each of which one workgroup solves alone.
constexpr uint32_t kRescueGrid = 64;

template <typename T, int NCT, int V, int WAVES, bool SYM = false, bool GVEC = false>
static hipError_t launch_fused_w(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s)
{
    const size_t lds = GVEC ? (size_t)align16<T>(2 * WAVES) * sizeof(T) : (size_t)FusedCarve<T>(a.n, a.N, WAVES, SYM).total * sizeof(T);
    if (lds > dev.lds_per_wg_max || (GVEC && !a.rescue_vec)) return hipErrorInvalidValue;
    auto kern = pcg_fused_kernel<T, NCT, V, WAVES, SYM, GVEC>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // persistent over problems: at most as many workgroups as can be resident
    uint32_t per_cu = (uint32_t)(dev.lds_per_cu / lds);
    const uint32_t by_waves = 32 / WAVES;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu == 0) per_cu = 1;
    uint32_t grid = (uint32_t)dev.num_cus * per_cu;
    static const int grid_cap = [] {
        const char *e = getenv("GBDPCG_FUSED_GRID");  // tuning runs only
        return e ? atoi(e) : 0;
    }();
    if (grid_cap > 0 && grid > (uint32_t)grid_cap) grid = (uint32_t)grid_cap;
    if (a.rescue && grid > kRescueGrid) grid = kRescueGrid;
    if (grid > a.batch) grid = a.batch;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
    return hipGetLastError();
}

template <typename T, int NCT, int V>
static hipError_t launch_fused_v(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s)
{
    // Few problems: give each the widest workgroup.  Many problems: 8-wave workgroups, two resident per
    // CU -- one streams while the other sits in its reduction / vector-update barriers, and only
    // 2 x CUs problems are in flight at once, which keeps their S + Pinv (re-read every iteration)
    // largely inside the 256 MiB Infinity Cache (measured on n=14, N=128, batch 1024: 4-wave
    // workgroups x 4 per CU 6.6 TB/s algorithmic, 8-wave x 2 per CU 7.7 TB/s; profiles/).
    // GBDPCG_FUSED_WAVES (4, 8 or 16) overrides the choice for tuning runs.
    static const int forced = [] {
        const char *e = getenv("GBDPCG_FUSED_WAVES");
        return e ? atoi(e) : 0;
    }();
    int waves = (a.batch < (uint32_t)dev.num_cus || a.rescue) ? 16 : 8;
    if (forced == 4 || forced == 8 || forced == 16) waves = forced;
    while (waves > 4 && fused_lds_bytes<T>(a.n, a.N, waves) > dev.lds_per_wg_max) waves /= 2;
    if constexpr (SymGeom<T, NCT>::OK) {
        const uintptr_t al = 2 * sizeof(T);
        const bool aligned = reinterpret_cast<uintptr_t>(a.S) % al == 0 &&
                             (!a.Pinv || reinterpret_cast<uintptr_t>(a.Pinv) % al == 0);
        if (a.symmetric && aligned && waves != 4) {
            if (waves == 16 && FusedCarve<T>(a.n, a.N, 16, true).total * sizeof(T) <= dev.lds_per_wg_max)
                return launch_fused_w<T, NCT, V, 16, true>(dev, a, s);
            if (FusedCarve<T>(a.n, a.N, 8, true).total * sizeof(T) <= dev.lds_per_wg_max)
                return launch_fused_w<T, NCT, V, 8, true>(dev, a, s);
        }
    }
    switch (waves) {
    case 16: return launch_fused_w<T, NCT, V, 16>(dev, a, s);
    case 8: return launch_fused_w<T, NCT, V, 8>(dev, a, s);
    default: return launch_fused_w<T, NCT, V, 4>(dev, a, s);
    }
}

template <typename T, int NCT>
static hipError_t launch_fused_n(const DeviceInfo &dev, const PcgArgs<T> &a, int V, hipStream_t s)
{
    if (V == 1) return launch_fused_v<T, NCT, 1>(dev, a, s);
    if constexpr (NCT == 0 || NCT % 2 == 0) {
        if (V == 2) return launch_fused_v<T, NCT, 2>(dev, a, s);
    }
    if constexpr (sizeof(T) == 4 && (NCT == 0 || NCT % 4 == 0)) {
        if (V == 4) return launch_fused_v<T, NCT, 4>(dev, a, s);
    }
    return hipErrorInvalidValue;
}

template <typename T> hipError_t launch_pcg_fused(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s)
{
    hipError_t rerr = hipSuccess;
    if (!a.rescue) {   // (a rescue launch streams: the kernels below are what it backs)
        if (launch_pcg_resident<T>(dev, a, s, &rerr)) return rerr;  // small problems: pcg_resident.hip
        if (a.symmetric && launch_pcg_resident_sym<T>(dev, a, s, &rerr)) return rerr;  // pcg_resident_sym.hip
        if (!a.symmetric && launch_pcg_cluster<T>(dev, a, s, &rerr)) {   // general storage over 2-4 CUs: pcg_cluster.hip
            // the workgroups of a cluster wait for each other: whatever they could not solve together is solved here
            return rerr == hipSuccess ? launch_pcg_rescue<T>(dev, a, s) : rerr;
        }
    }
    const void *ptrs[] = {a.S, a.Pinv};
    const int V = choose_vec<T>(a.n, ptrs, 2);
    if (V == 0) return hipErrorInvalidValue;
    static const bool generic_only = getenv("GBDPCG_FORCE_GENERIC") != nullptr;  // tuning runs only
    if (!generic_only) {
#define GBDPCG_CASE(NN) \
    if (a.n == NN && V == best_v<T, NN>()) return launch_fused_v<T, NN, best_v<T, NN>()>(dev, a, s);
        GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    }
    return launch_fused_n<T, 0>(dev, a, V, s);
}

template <typename T> size_t rescue_vec_bytes(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch)
{
    if (fused_fits<T>(dev, n, N)) return 0;
    const uint32_t grid = batch < kRescueGrid ? batch : kRescueGrid;
    return (size_t)grid * FusedCarve<T>(n, N, 16).red * sizeof(T);
}

template <typename T> hipError_t launch_pcg_rescue(const DeviceInfo &dev, PcgArgs<T> a, hipStream_t s)
{
#ifdef GBDPCG_TEST_HOOKS
    // variants/libgbdpcg_hooks.so only: let a test see what the kernel that gave up left behind
    if (getenv("GBDPCG_RESCUE_OFF")) return hipSuccess;
#endif
    a.rescue = true;
    a.symmetric = false;      // general storage: the kernel reads L, D and R whatever the launch it backs assumed
    a.host_done = nullptr;
    if (fused_fits<T>(dev, a.n, a.N)) return launch_pcg_fused<T>(dev, a, s);
    // vectors in device memory, runtime-n kernel (the block sizes that reach this are the persistent path's)
    const void *ptrs[] = {a.S, a.Pinv};
    const int V = choose_vec<T>(a.n, ptrs, 2);
    if (V == 1) return launch_fused_w<T, 0, 1, 16, false, true>(dev, a, s);
    if (V == 2) return launch_fused_w<T, 0, 2, 16, false, true>(dev, a, s);
    if constexpr (sizeof(T) == 4) {
        if (V == 4) return launch_fused_w<T, 0, 4, 16, false, true>(dev, a, s);
    }
    return hipErrorInvalidValue;
}

template bool fused_has_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);
template bool fused_has_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);
template size_t fused_lds_bytes<float>(uint32_t, uint32_t, uint32_t);
template size_t fused_lds_bytes<double>(uint32_t, uint32_t, uint32_t);
template bool fused_fits<float>(const DeviceInfo &, uint32_t, uint32_t);
template bool fused_fits<double>(const DeviceInfo &, uint32_t, uint32_t);
template hipError_t launch_pcg_fused<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t);
template hipError_t launch_pcg_fused<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t);
template hipError_t launch_pcg_rescue<float>(const DeviceInfo &, PcgArgs<float>, hipStream_t);
template hipError_t launch_pcg_rescue<double>(const DeviceInfo &, PcgArgs<double>, hipStream_t);
template size_t rescue_vec_bytes<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);
template size_t rescue_vec_bytes<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);

}  // namespace gbdpcg
