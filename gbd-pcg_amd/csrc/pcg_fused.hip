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

#include "pcg_stream.hpp"

namespace gbdpcg {

// SYM: both matrices are symmetric block-tridiagonal (L_{k+1} == R_k^T) and are streamed through
// SymStream (bt_sym.hpp): [D_k | R_k] only, 2/3 of the bytes.  The per-problem solve is StreamSolver::solve (pcg_stream.hpp).
template <typename T, int NCT, int V, int WAVES, bool SYM>
__global__ __launch_bounds__(WAVES * 64) void pcg_fused_kernel(PcgArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    const uint32_t n = NCT ? (uint32_t)NCT : a.n;
    const FusedCarve<T> cv(n, a.N, WAVES, SYM);
    StreamSolver<T, NCT, V, WAVES, SYM> sv(n, threadIdx.x);
    for (uint32_t prob = blockIdx.x; prob < a.batch; prob += gridDim.x) {
        if (!pcg_takes(a, prob)) continue;  // this launch is not the one that owns the problem
        sv.solve(a, prob, smem + cv.xa, smem + cv.xb, smem + cv.yc, smem + cv.lam, smem + cv.red, smem + cv.zc);
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

template <typename T, int NCT, int V, int WAVES, bool SYM = false>
static hipError_t launch_fused_w(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s)
{
    const size_t lds = (size_t)FusedCarve<T>(a.n, a.N, WAVES, SYM).total * sizeof(T);
    if (lds > dev.lds_per_wg_max) return hipErrorInvalidValue;
    auto kern = pcg_fused_kernel<T, NCT, V, WAVES, SYM>;
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
    int waves = a.batch < (uint32_t)dev.num_cus ? 16 : 8;
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
    if (launch_pcg_resident<T>(dev, a, s, &rerr)) return rerr;  // small problems: pcg_resident.hip
    if (a.symmetric && launch_pcg_resident_sym<T>(dev, a, s, &rerr)) return rerr;  // pcg_resident_sym.hip
    if (!a.symmetric && launch_pcg_cluster<T>(dev, a, s, &rerr)) return rerr;      // general storage over 2-4 CUs: pcg_cluster.hip
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

template bool fused_has_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);
template bool fused_has_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t);
template size_t fused_lds_bytes<float>(uint32_t, uint32_t, uint32_t);
template size_t fused_lds_bytes<double>(uint32_t, uint32_t, uint32_t);
template bool fused_fits<float>(const DeviceInfo &, uint32_t, uint32_t);
template bool fused_fits<double>(const DeviceInfo &, uint32_t, uint32_t);
template hipError_t launch_pcg_fused<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t);
template hipError_t launch_pcg_fused<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t);

}  // namespace gbdpcg
