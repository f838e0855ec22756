// pcg_resident.hip -- PCG with BOTH matrices register-resident: one 8-wave workgroup per problem.
//
// Replaces pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218) for small problems (BASELINE config 2:
// n = 14, N = 64, fp32).  See the comment on the lane map below; the iteration itself is
// pcg.cuh:118-208 with every vector element living in the register of the lane that owns its row:
// lambda, r, p, S p and Pinv r never touch LDS; only p and r are mirrored into LDS (padded) because
// the neighbouring block-rows read them as x operands.
#include <cstdlib>

#include "bt_dense.hpp"
#include "bt_device.hpp"
#include "internal.hpp"

#ifndef GBDPCG_RES_OCC
#define GBDPCG_RES_OCC 1   // 0: no register cap on the small-block instantiations of the resident kernel (A/B runs)
#endif
#if GBDPCG_RES_OCC
#define GBDPCG_RES_OCC_ATTR __attribute__((amdgpu_waves_per_eu(NCT * sizeof(T) <= 32 || (NCT <= 6) || (V == 1 && sizeof(T) == 4 && NCT <= 11) ? 4 : 2)))   // n <= 8 in fp32, n <= 6 in fp64, odd n <= 11 in fp32 (one row per lane: 6n matrix registers)
#else
#define GBDPCG_RES_OCC_ATTR
#endif

namespace gbdpcg {

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

// STAGED (fp32, n = 14, matrices 16-byte aligned): the tiles come in through LDS-DMA staging buffers -- dense, coalesced
// 16-byte pieces instead of 8 bytes per lane at a 56-byte stride (bt_dense.hpp, dense_staged_load).
// (Blocks of 8 or fewer -- 6 in fp64, where n = 8 would spill 40 registers -- are held to 128 registers, so that two workgroups
// share a compute unit: 1024 converged solves of n = 8, N = 128 take 105 instead of 119 us, n = 6, N = 80 in fp64 92 instead of 120.)
template <typename T, int NCT, int V, bool STAGED = false>
__global__ __launch_bounds__(512) GBDPCG_RES_OCC_ATTR void pcg_resident_kernel(PcgArgs<T> a)
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
    const uint32_t stage_offset_floats = align16<float>(2 * padded + 2 * align16<T>(WAVES));   // STAGED only (T = float)
    (void)stage_offset_floats;
    const DenseCtx<T, NCT, V> dc(wave, lane, N);
    const uint32_t row0 = (dc.live ? dc.kl : 0u) * n + dc.rp * V;  // first of this lane's rows
    const size_t mstride = (size_t)3 * n * n * N;

    for (uint32_t prob = blockIdx.x; prob < a.batch; prob += gridDim.x) {
        const T *S = a.S + prob * mstride;
        const T *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;
        const size_t voff = (size_t)prob * len;

        DenseTile<T, NCT, V> tS, tP;
        if constexpr (STAGED) {
            // the staging buffers sit behind the mirrors and the reduction words in dynamic LDS (16-byte aligned)
            float *stage = reinterpret_cast<float *>(smem_raw) + stage_offset_floats;
            dense_staged_load<T, NCT, V>(S, P, N, dc, wave, lane, 0u, N, reinterpret_cast<unsigned char *>(stage), tS, tP, [] {});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            dense_load<T, NCT, V>(S, N, dc, tS);
            if (P) dense_load<T, NCT, V>(P, N, dc, tP);
        }

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

// Taken whenever the shape fits: the matrices are then read once per solve instead of once per iteration.  Built for the even
// block sizes below (fp32: two rows per lane; fp64: one row per lane, not for n = 14 -- 168 matrix VGPRs plus the fp64 working
// set spill): n = 14 is BASELINE config 2 (N <= 72); the smaller blocks (stateSize of 2-6 joint arms) fit longer horizons
// (n = 8: N <= 128, n = 10: N <= 96, n = 4: N <= 256 in fp32) and leave room for several workgroups per compute unit.
// GBDPCG_NO_RESIDENT disables the path (tuning runs).
// n = 2 (the pendulum, and the block size of the reference's own example) from 16 knots on: 857 -> 52 us per 1024 converged solves
// of N = 128.  Shorter problems of that block size stay with the streaming kernel: nothing is gained on 32 rows, and the
// equal-iteration-count pin of the reference's example system (n = 2, N = 3, kappa ~ 1562, fp32: the count depends on the
// summation order, and this kernel's order meets the exit test one iteration earlier) was taken with that kernel's order.
#define GBDPCG_RESIDENT_N(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// rows per lane: two in fp32 where the block size is even (a lane's rows of a column are one 8-byte load), else one
template <typename T, int NN> constexpr int resident_rows() { return sizeof(T) == 4 && NN % 2 == 0 ? 2 : 1; }
// ... and whether the kernel is built for the pair at all: both tiles must leave registers for the solve (fp64 at 14 and 15: 168 / 180
// matrix registers plus the fp64 working set spill -- the cluster kernel has those, with part of Pinv in LDS)
template <typename T, int NN> constexpr bool resident_built()
{
    return 2 * DenseGeom<T, NN, resident_rows<T, NN>()>::REGS <= 176 && !(sizeof(T) == 8 && NN == 14);
}
constexpr uint32_t kResidentMinKnotsN2 = 16;

template <typename T> bool resident_shape(uint32_t n, uint32_t N)
{
    static const bool off = getenv("GBDPCG_NO_RESIDENT") != nullptr;
    if (off) return false;
#define GBDPCG_X(NN) \
    if (n == NN) return resident_built<T, NN>() && !(NN == 2 && N < kResidentMinKnotsN2) && N <= DenseGeom<T, NN, resident_rows<T, NN>()>::MAX_KNOTS;
    GBDPCG_RESIDENT_N(GBDPCG_X)
#undef GBDPCG_X
    return false;
}

// Workgroups of `kern` one compute unit holds at once (registers, LDS); asked once per kernel and LDS size class.
template <typename K> static uint32_t resident_wgs_per_cu(K kern, uint32_t threads, size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, (int)threads, lds) != hipSuccess || nb < 1) nb = 1;
    return (uint32_t)(nb > 4 ? 4 : nb);
}

// Workgroups per compute unit of pcg_resident_kernel<T, NN, RV> at this LDS size: one count per LDS size class of 8 KB (the
// kernel's registers do not depend on the horizon), asked of the runtime once -- never while a graph is being captured
// (query = false: an unknown class then counts as 1; gbdpcg_graph_create_* asks before it starts the capture).
template <typename T, int NN> static uint32_t resident_per_cu(size_t lds, bool query)
{
    constexpr int RV = resident_rows<T, NN>();
    static uint32_t per_cu[32] = {};
    const uint32_t cls = (uint32_t)(lds >> 13) < 31u ? (uint32_t)(lds >> 13) : 31u;
    uint32_t w = __atomic_load_n(&per_cu[cls], __ATOMIC_RELAXED);
    if (!w && query) {
        w = resident_wgs_per_cu(pcg_resident_kernel<T, NN, RV>, DenseGeom<T, NN, RV>::WAVES * 64, ((size_t)cls + 1) << 13);
        __atomic_store_n(&per_cu[cls], w, __ATOMIC_RELAXED);
    }
    return w ? w : 1u;
}

template <typename T> void resident_prepare(uint32_t n, uint32_t N)
{
    if (!resident_shape<T>(n, N)) return;
    const size_t lds = ((size_t)2 * align16<T>((N + 2) * n) + 2 * align16<T>(8)) * sizeof(T);
#define GBDPCG_X(NN)                                     \
    if constexpr (resident_built<T, NN>()) {             \
        if (n == NN) (void)resident_per_cu<T, NN>(lds, true); \
    }
    GBDPCG_RESIDENT_N(GBDPCG_X)
#undef GBDPCG_X
}

template <typename T, int NN>
static bool launch_pcg_resident_n(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err)
{
    constexpr int RV = resident_rows<T, NN>();
    using Dg = DenseGeom<T, NN, RV>;
    static_assert(2 * Dg::REGS <= 176, "resident matrices must leave registers for the solve");
    const uintptr_t al = RV * sizeof(T);
    if ((reinterpret_cast<uintptr_t>(a.S) % al) || (a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % al)) return false;
    size_t lds = ((size_t)2 * align16<T>((a.N + 2) * a.n) + 2 * align16<T>(Dg::WAVES)) * sizeof(T);
    if constexpr (sizeof(T) == 4 && NN == 14) {
        static const bool no_staging = getenv("GBDPCG_RES_DIRECT_LOADS") != nullptr;   // tuning runs only
        const bool staged = !no_staging && !(reinterpret_cast<uintptr_t>(a.S) % 16) && !(a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % 16);
        if (staged) {
            uint32_t grid = (uint32_t)dev.num_cus;  // one resident workgroup owns a CU's register file
            if (grid > a.batch) grid = a.batch;
            lds = (lds + 15) / 16 * 16 + dense_stage_lds_bytes<T, 14, RV>();
            auto kern = pcg_resident_kernel<T, 14, RV, true>;
            *err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (*err != hipSuccess) return true;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(Dg::WAVES * 64), lds, s, a);
            *err = hipGetLastError();
            return true;
        }
    }
    auto kern = pcg_resident_kernel<T, NN, RV>;
    // the smaller blocks need few registers: several workgroups share a compute unit and fill each other's barrier waits
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    const uint32_t w = resident_per_cu<T, NN>(lds, !capturing);
    uint64_t grid = (uint64_t)dev.num_cus * w;
    if (grid > a.batch) grid = a.batch;
    if (lds > 48 * 1024) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (*err != hipSuccess) return true;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(Dg::WAVES * 64), lds, s, a);
    *err = hipGetLastError();
    return true;
}

template <typename T>
bool launch_pcg_resident(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err)
{
    if (!resident_shape<T>(a.n, a.N)) return false;
#define GBDPCG_X(NN)                                                        \
    if constexpr (resident_built<T, NN>()) {                                \
        if (a.n == NN) return launch_pcg_resident_n<T, NN>(dev, a, s, err); \
    }
    GBDPCG_RESIDENT_N(GBDPCG_X)
#undef GBDPCG_X
    return false;
}

template void resident_prepare<float>(uint32_t, uint32_t);
template void resident_prepare<double>(uint32_t, uint32_t);
template bool resident_shape<float>(uint32_t, uint32_t);
template bool resident_shape<double>(uint32_t, uint32_t);
template bool launch_pcg_resident<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t, hipError_t *);
template bool launch_pcg_resident<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t, hipError_t *);

}  // namespace gbdpcg
