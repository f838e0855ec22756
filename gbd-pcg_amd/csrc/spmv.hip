// spmv.hip -- batched block-tridiagonal SpMV  y = M x  (the HBM-roofline kernel).
//
// Replaces loadbdVec + bdmv (/root/reference/include/utils.cuh:9-85) as a standalone operator.
// Work decomposition: a workgroup owns `rpw` consecutive block-rows of one problem; it stages the
// (rpw+2)*n halo window of x in LDS once, then each of its wavefronts streams whole block-rows
// from HBM with RowStream (bt_device.hpp).  Algorithmic bytes per problem:
// ((3N-2) n^2 + 2 n N) sizeof(T)  (SURVEY.md section 8d).
#include <cstdlib>

#include "bt_device.hpp"
#include "bt_sym.hpp"
#include "internal.hpp"

namespace gbdpcg {

// The product reads every matrix byte exactly once: stream it with the non-temporal policy.
#ifndef GBDPCG_SPMV_NT
#define GBDPCG_SPMV_NT 1
#endif
constexpr bool kSpmvNT = GBDPCG_SPMV_NT != 0;

template <typename T, int NCT, int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmv_kernel(SpmvArgs<T> a, uint32_t rpw, uint32_t chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw);

    const uint32_t n = NCT ? (uint32_t)NCT : a.n;
    const uint32_t N = a.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps row bases in SGPRs
    const uint32_t prob = blockIdx.x / chunks;
    const uint32_t chunk_id = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk_id * rpw;
    const uint32_t k1 = min(N, k0 + rpw);
    const size_t len = (size_t)n * N;

    // matrix loads first: they do not depend on x, so they are in flight while x is staged
    const LaneMap<NCT, V> m(n, lane);
    const StreamCtx<T, NCT, V> cx(m, lane);
    const T *M = a.M + (size_t)prob * 3 * n * n * N;
    RowStream<T, NCT, V, kSpmvNT> rs;
    rs.prime(M, k0 + wave, k1, WAVES, cx, n);

    // halo window [x_{k0-1} .. x_{k1}] with zeros outside the vector
    const T *x = a.x + (size_t)prob * len;
    const uint32_t cnt = (k1 - k0 + 2) * n;
    const int64_t g0 = (int64_t)k0 * n - n;
    for (uint32_t i = tid; i < cnt; i += WAVES * 64) {
        const int64_t gi = g0 + i;
        xs[i] = (gi >= 0 && gi < (int64_t)len) ? x[gi] : T(0);
    }
    __syncthreads();

    // y of this chunk is collected in LDS (no global stores inside the streaming loop) and written
    // out as one dense run at the end
    T *y = a.y + (size_t)prob * len;
    T *ys = xs + align16<T>((rpw + 2) * n);
    rs.run(xs, k0, N, m, cx, lane, [&](uint32_t k, const T(&acc)[V]) __attribute__((always_inline)) {
        if (m.active && m.g == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) ys[(k - k0) * n + m.rp * V + v] = acc[v];
        }
    });
    __syncthreads();
    const uint32_t ycnt = (k1 - k0) * n;
    for (uint32_t i = tid; i < ycnt; i += WAVES * 64) y[(size_t)k0 * n + i] = ys[i];
}

// Symmetric storage (gbdpcg_set_symmetric(1)): only [D_k | R_k] is read.  A workgroup owns the rows
// [k0, k1) and additionally streams the row before its chunk for that row's transposed product
// R_{k0-1}^T x_{k0-1}, which lands in y_{k0}; the transposed product of its own last row belongs to
// the next chunk and is dropped.  One extra block-row per chunk (1.5 % at 64 rows per workgroup).
template <typename T, int NCT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmv_sym_kernel(SpmvArgs<T> a, uint32_t rpw, uint32_t chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr uint32_t n = SymGeom<T, NCT>::N_;
    T *xs = reinterpret_cast<T *>(smem_raw);           // [x_{k0-1} .. x_{k1}]
    T *ys = xs + align16<T>((rpw + 2) * n);            // D_k x_k + R_k x_{k+1}
    T *zs = ys + align16<T>(rpw * n);                  // R_{k-1}^T x_{k-1}
    const uint32_t N = a.N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t prob = blockIdx.x / chunks;
    const uint32_t chunk_id = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk_id * rpw, k1 = min(N, k0 + rpw);
    const size_t len = (size_t)n * N;

    const SymCtx<T, NCT> cx(lane);
    SymStream<T, NCT, kSpmvNT> ss;
    const uint32_t first = k0 ? k0 - 1 : 0u;
    ss.prime(a.M + (size_t)prob * 3 * n * n * N, first + wave, k1, WAVES, cx);

    const T *x = a.x + (size_t)prob * len;
    const uint32_t cnt = (k1 - k0 + 2) * n;
    const int64_t g0 = (int64_t)k0 * n - n;
    for (uint32_t i = tid; i < cnt; i += WAVES * 64) {
        const int64_t gi = g0 + i;
        xs[i] = (gi >= 0 && gi < (int64_t)len) ? x[gi] : T(0);
    }
    if (k0 == 0)
        for (uint32_t i = tid; i < n; i += WAVES * 64) zs[i] = T(0);  // row 0 has no block-row above it
    __syncthreads();

    // x_k sits at xs + n + (k - k0) n
    ss.run(xs + n - (size_t)k0 * n, N, cx,
           [&](uint32_t k, T a0, T a1) __attribute__((always_inline)) {
               if (k >= k0 && cx.g == 0 && cx.act) {
                   ys[(k - k0) * n + cx.rp * 2] = a0;
                   ys[(k - k0) * n + cx.rp * 2 + 1] = a1;
               }
           },
           [&](uint32_t k, uint32_t c, T t) __attribute__((always_inline)) {
               if (k + 1 < k1 && cx.rp == 0) zs[(k + 1 - k0) * n + c - n] = t;
           });
    __syncthreads();
    T *y = a.y + (size_t)prob * len;
    const uint32_t ycnt = (k1 - k0) * n;
    for (uint32_t i = tid; i < ycnt; i += WAVES * 64) y[(size_t)k0 * n + i] = ys[i] + zs[i];
}

template <typename T, int NCT>
static hipError_t launch_spmv_sym(const DeviceInfo &dev, const SpmvArgs<T> &a, hipStream_t s)
{
    constexpr int WAVES = 4;
    const uint64_t total_rows = (uint64_t)a.N * a.batch;
    const uint64_t target_wgs = (uint64_t)dev.num_cus * 8;
    uint32_t rpw = (uint32_t)((total_rows + target_wgs - 1) / target_wgs);
    rpw = (rpw + WAVES - 1) / WAVES * WAVES;
    if (rpw > a.N) rpw = a.N;
    if (rpw == 0) rpw = 1;
    const uint32_t chunks = (a.N + rpw - 1) / rpw;
    const size_t lds = ((size_t)align16<T>((rpw + 2) * a.n) + 2 * (size_t)align16<T>(rpw * a.n)) * sizeof(T);
    if (lds > dev.lds_per_wg_max) return hipErrorInvalidValue;
    auto kern = spmv_sym_kernel<T, NCT, WAVES>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(chunks * a.batch), dim3(WAVES * 64), lds, s, a, rpw, chunks);
    return hipGetLastError();
}

template <typename T, int NCT, int V>
static hipError_t launch_spmv_v(const DeviceInfo &dev, const SpmvArgs<T> &a, hipStream_t s)
{
    constexpr int WAVES = 4;
    // ~8 workgroups per CU; every wave gets whole block-rows
    const uint64_t total_rows = (uint64_t)a.N * a.batch;
    const uint64_t target_wgs = (uint64_t)dev.num_cus * 8;
    uint32_t rpw = (uint32_t)((total_rows + target_wgs - 1) / target_wgs);
    rpw = (rpw + WAVES - 1) / WAVES * WAVES;
    if (rpw > a.N) rpw = a.N;
    if (rpw == 0) rpw = 1;
    const uint32_t chunks = (a.N + rpw - 1) / rpw;
    const size_t lds = ((size_t)align16<T>((rpw + 2) * a.n) + align16<T>(rpw * a.n)) * sizeof(T);
    if (lds > dev.lds_per_wg_max) return hipErrorInvalidValue;
    auto kern = spmv_kernel<T, NCT, V, WAVES>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(chunks * a.batch), dim3(WAVES * 64), lds, s, a, rpw, chunks);
    return hipGetLastError();
}

template <typename T, int NCT>
static hipError_t launch_spmv_n(const DeviceInfo &dev, const SpmvArgs<T> &a, int V, hipStream_t s)
{
    // only instantiate (NCT, V) pairs choose_vec can produce
    if (V == 1) return launch_spmv_v<T, NCT, 1>(dev, a, s);
    if constexpr (NCT == 0 || NCT % 2 == 0) {
        if (V == 2) return launch_spmv_v<T, NCT, 2>(dev, a, s);
    }
    if constexpr (sizeof(T) == 4 && (NCT == 0 || NCT % 4 == 0)) {
        if (V == 4) return launch_spmv_v<T, NCT, 4>(dev, a, s);
    }
    return hipErrorInvalidValue;
}

template <typename T> hipError_t launch_spmv(const DeviceInfo &dev, const SpmvArgs<T> &a, hipStream_t s)
{
    const void *ptrs[] = {a.M};
    const int V = choose_vec<T>(a.n, ptrs, 1);
    if (V == 0) return hipErrorInvalidValue;
    static const bool generic_only = getenv("GBDPCG_FORCE_GENERIC") != nullptr;  // tuning runs only
    if (a.symmetric && reinterpret_cast<uintptr_t>(a.M) % (2 * sizeof(T)) == 0) {
#define GBDPCG_CASE(NN) \
    if constexpr (SymGeom<T, NN>::OK) { if (a.n == NN) return launch_spmv_sym<T, NN>(dev, a, s); }
        GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    }
    if (!generic_only) {
#define GBDPCG_CASE(NN) \
    if (a.n == NN && V == best_v<T, NN>()) return launch_spmv_v<T, NN, best_v<T, NN>()>(dev, a, s);
        GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    }
    return launch_spmv_n<T, 0>(dev, a, V, s);
}

template hipError_t launch_spmv<float>(const DeviceInfo &, const SpmvArgs<float> &, hipStream_t);
template hipError_t launch_spmv<double>(const DeviceInfo &, const SpmvArgs<double> &, hipStream_t);

}  // namespace gbdpcg
