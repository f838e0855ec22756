// pcg_resident.hip -- PCG with BOTH matrices register-resident: one 8-wave workgroup per problem.
//
// Replaces pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218) for small problems (BASELINE config 2:
// n = 14, N = 64, fp32).  See the comment on the lane map below; the iteration itself is
// pcg.cuh:118-208 with every vector element living in the register of the lane that owns its row:
// lambda, r, p, S p and Pinv r never touch LDS; only p and r are mirrored into LDS (padded) because
// the neighbouring block-rows read them as x operands.
#include <cstdlib>

#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

// When both matrices of a problem fit the register file of ONE 8-wave workgroup they are read from
// HBM exactly once and an iteration touches only LDS.  This is the reference's own idea (its
// block-rows sit in shared memory for the whole solve, pcg.cuh:104-110) moved one level up the
// hierarchy, with the whole problem inside one workgroup so that no grid barrier exists.
//
// Lane map (different from the streaming one, where every instruction must be a dense global read):
// a lane owns V whole ROWS of one block-row -- all 3n columns, 3n*V registers per matrix -- so a
// block-row product is 3n FMAs per row with NO cross-lane fold, columns accumulated in ascending
// order exactly like bdmv (utils.cuh:77-81).  n/V lanes make a block-row, BPW = 64/(n/V) block-rows
// make a wave (n=14, V=2: 7 lanes x 9 block-rows = 63 lanes; the reference keeps n of 64 threads
// busy), 8 waves cover N <= 8*BPW knots.  x is read from LDS as 8/16-byte pairs shared by the lanes
// of a block-row.  Edge columns (L_0, R_{N-1}) and rows past N are zeroed once, at load time.
template <typename T, int NCT, int V> struct DenseGeom {
    static constexpr uint32_t N_ = NCT > 0 ? NCT : 2;
    static constexpr uint32_t LPB = N_ / V > 0 ? N_ / V : 1;  // lanes per block-row
    static constexpr uint32_t BPW = kWave / LPB;        // block-rows per wave
    static constexpr uint32_t COLS = 3 * N_;
    static constexpr uint32_t REGS = COLS * V * sizeof(T) / 4;  // VGPRs per lane per matrix
    static constexpr uint32_t WAVES = 8;
    static constexpr uint32_t MAX_KNOTS = WAVES * BPW;
};

template <typename T, int NCT, int V> struct DenseTile {
    T a[DenseGeom<T, NCT, V>::COLS][V];
};

template <typename T, int NCT, int V> struct DenseCtx {
    uint32_t k;       // this lane's block-row
    uint32_t rp;      // row group inside it
    bool live;        // lane maps to a real row of a real block-row
    __device__ __forceinline__ DenseCtx(uint32_t wave, uint32_t lane, uint32_t N) {
        using Dg = DenseGeom<T, NCT, V>;
        const uint32_t b = lane / Dg::LPB;
        rp = lane - b * Dg::LPB;
        k = wave * Dg::BPW + b;
        live = b < Dg::BPW && k < N;
    }
};

template <typename T, int NCT, int V>
__device__ __forceinline__ void dense_load(const T *__restrict__ M, uint32_t N, const DenseCtx<T, NCT, V> &dc,
                                           DenseTile<T, NCT, V> &tl)
{
    using Dg = DenseGeom<T, NCT, V>;
    const uint32_t k = dc.live ? dc.k : 0u;
    const T *src = M + (size_t)k * 3 * Dg::N_ * Dg::N_ + dc.rp * V;
    const uint32_t c_lo = dc.k == 0 ? Dg::N_ : 0u, c_hi = dc.k == N - 1 ? 2 * Dg::N_ : 3 * Dg::N_;
#pragma unroll
    for (uint32_t c = 0; c < Dg::COLS; ++c) {
        T v[V];
        VecIO<T, V>::load(src + c * Dg::N_, v);
        const bool keep = dc.live && c >= c_lo && c < c_hi;
#pragma unroll
        for (int j = 0; j < V; ++j) tl.a[c][j] = keep ? v[j] : T(0);
    }
}

// y_k = [L|D|R]_k * X-window for this lane's rows, left in registers.
template <typename T, int NCT, int V>
__device__ __forceinline__ void dense_mv(const DenseTile<T, NCT, V> &tl, const T *X, const DenseCtx<T, NCT, V> &dc,
                                         T (&acc)[V])
{
    using Dg = DenseGeom<T, NCT, V>;
    using P2 = typename VecOf<T, 2>::type;
    const uint32_t k = dc.live ? dc.k : 0u;
    const P2 *xk = reinterpret_cast<const P2 *>(X + k * Dg::N_);  // column c of row k multiplies X[k*n + c]
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = T(0);
#pragma unroll
    for (uint32_t c = 0; c < Dg::COLS; c += 2) {
        const P2 xv = xk[c / 2];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fma_t(tl.a[c][j], xv.x, acc[j]);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fma_t(tl.a[c + 1][j], xv.y, acc[j]);
    }
}

// Workgroup-wide sum of per-lane partials; every thread returns the same bits (one barrier inside).
template <typename T, int WAVES>
__device__ __forceinline__ T wg_sum_r(T part, T *red, uint32_t lane, uint32_t wave)
{
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    T tot = red[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) tot += red[w];
    return tot;
}

template <typename T, int NCT, int V>
__global__ __launch_bounds__(512) void pcg_resident_kernel(PcgArgs<T> a)
{
    using Dg = DenseGeom<T, NCT, V>;
    constexpr int WAVES = Dg::WAVES;
    static_assert(WAVES * 64 == 512, "launch bounds above assume 8 waves");
    constexpr uint32_t THREADS = WAVES * 64, n = Dg::N_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);

    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t padded = align16<T>((N + 2) * n);
    T *xa = smem, *xb = xa + padded;          // padded mirrors of p (lambda in the prologue) and r
    T *red0 = xb + padded, *red1 = red0 + WAVES;
    const DenseCtx<T, NCT, V> dc(wave, lane, N);
    const uint32_t row0 = (dc.live ? dc.k : 0u) * n + dc.rp * V;  // first of this lane's rows
    const size_t mstride = (size_t)3 * n * n * N;

    for (uint32_t prob = blockIdx.x; prob < a.batch; prob += gridDim.x) {
        const T *S = a.S + prob * mstride;
        const T *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;
        const size_t voff = (size_t)prob * len;

        DenseTile<T, NCT, V> tS, tP;
        dense_load<T, NCT, V>(S, N, dc, tS);
        if (P) dense_load<T, NCT, V>(P, N, dc, tP);

        T lamv[V], rv[V], pv[V], yv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) lamv[j] = dc.live ? a.lambda[voff + row0 + j] : T(0);
        for (uint32_t i = tid; i < n; i += THREADS) {
            xa[i] = T(0); xa[n + len + i] = T(0);
            xb[i] = T(0); xb[n + len + i] = T(0);
        }
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xa[n + row0 + j] = lamv[j];
        }
        __syncthreads();

        // r = gamma - S lambda                                            (pcg.cuh:118-126)
        dense_mv<T, NCT, V>(tS, xa, dc, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) rv[j] = dc.live ? a.gamma[voff + row0 + j] - yv[j] : T(0);
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xb[n + row0 + j] = rv[j];
        }
        __syncthreads();

        // r~ = Pinv r ; p = r~ ; eta = r.r~                               (pcg.cuh:130-149)
        T part = T(0);
        if (P) dense_mv<T, NCT, V>(tP, xb, dc, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            pv[j] = P ? yv[j] : rv[j];
            part = fma_t(rv[j], pv[j], part);
        }
        T eta = wg_sum_r<T, WAVES>(part, red1, lane, wave);
        if (dc.live) {  // every read of xa (as lambda) happened before the barrier inside wg_sum_r
#pragma unroll
            for (int j = 0; j < V; ++j) xa[n + row0 + j] = pv[j];
        }
        __syncthreads();

        uint32_t iter = 0;
        bool max_iter_exit = true;
        for (; iter < a.max_iter; ++iter) {                               // pcg.cuh:154
            // upsilon = S p ; alpha = eta / (p.upsilon)                   (pcg.cuh:156-169)
            dense_mv<T, NCT, V>(tS, xa, dc, yv);
            part = T(0);
#pragma unroll
            for (int j = 0; j < V; ++j) part = fma_t(pv[j], yv[j], part);
            const T alpha = eta / wg_sum_r<T, WAVES>(part, red0, lane, wave);
            // lambda += alpha p ; r -= alpha upsilon                      (pcg.cuh:172-176)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                lamv[j] = fma_t(alpha, pv[j], lamv[j]);
                rv[j] = fma_t(-alpha, yv[j], rv[j]);
            }
            if (dc.live) {
#pragma unroll
                for (int j = 0; j < V; ++j) xb[n + row0 + j] = rv[j];
            }
            __syncthreads();
            // r~ = Pinv r ; eta_new = r.r~                                (pcg.cuh:180-193)
            if (P) dense_mv<T, NCT, V>(tP, xb, dc, yv);
            part = T(0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (!P) yv[j] = rv[j];
                part = fma_t(rv[j], yv[j], part);
            }
            const T eta_new = wg_sum_r<T, WAVES>(part, red1, lane, wave);
            if (fabs(eta_new) < a.tol) {                                  // pcg.cuh:195
                ++iter;
                max_iter_exit = false;
                break;
            }
            const T beta = eta_new / eta;                                 // pcg.cuh:199-206
            eta = eta_new;
#pragma unroll
            for (int j = 0; j < V; ++j) pv[j] = fma_t(beta, pv[j], yv[j]);
            if (dc.live) {
#pragma unroll
                for (int j = 0; j < V; ++j) xa[n + row0 + j] = pv[j];
            }
            __syncthreads();
        }

        // outputs                                                         (pcg.cuh:212,215)
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                a.lambda[voff + row0 + j] = lamv[j];
                if (a.r) a.r[voff + row0 + j] = rv[j];
                if (a.p) a.p[voff + row0 + j] = pv[j];
            }
        }
        if (tid == 0) {
            a.iters[prob] = iter;
            if (a.max_iter_exit) a.max_iter_exit[prob] = max_iter_exit ? 1 : 0;
        }
        __syncthreads();  // LDS mirrors are reused by the next problem
    }
}

// Taken whenever the shape fits (n = 14, fp32, N <= 72): the matrices are then read once per solve
// instead of once per iteration.  The fp64 map (1 row per lane, N <= 32) is written but disabled: 168
// matrix VGPRs plus the fp64 working set spill.  GBDPCG_NO_RESIDENT disables the path (tuning runs).
template <typename T> bool resident_shape(uint32_t n, uint32_t N)
{
    static const bool off = getenv("GBDPCG_NO_RESIDENT") != nullptr;
    if (off || n != 14 || sizeof(T) == 8) return false;
    return N <= DenseGeom<T, 14, (sizeof(T) == 4 ? 2 : 1)>::MAX_KNOTS;
}

template <typename T>
bool launch_pcg_resident(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err)
{
    if (!resident_shape<T>(a.n, a.N)) return false;
    constexpr int RV = sizeof(T) == 4 ? 2 : 1;
    using Dg = DenseGeom<T, 14, RV>;
    static_assert(2 * Dg::REGS <= 176, "resident matrices must leave registers for the solve");
    const uintptr_t al = RV * sizeof(T);
    if ((reinterpret_cast<uintptr_t>(a.S) % al) || (a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % al)) return false;
    const size_t lds = ((size_t)2 * align16<T>((a.N + 2) * a.n) + 2 * align16<T>(Dg::WAVES)) * sizeof(T);
    uint32_t grid = (uint32_t)dev.num_cus;  // one resident workgroup owns a CU's register file
    if (grid > a.batch) grid = a.batch;
    hipLaunchKernelGGL((pcg_resident_kernel<T, 14, RV>), dim3(grid), dim3(Dg::WAVES * 64), lds, s, a);
    *err = hipGetLastError();
    return true;
}

template bool resident_shape<float>(uint32_t, uint32_t);
template bool resident_shape<double>(uint32_t, uint32_t);
template bool launch_pcg_resident<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t, hipError_t *);
template bool launch_pcg_resident<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t, hipError_t *);

}  // namespace gbdpcg
