// pcg_persist.hip -- PCG for ONE large problem spread over many CUs inside a single persistent launch: the host side, and
// the kernels of the first group of block sizes.  The kernels themselves (and what they do) are in pcg_persist_kernels.hpp;
// their instantiations -- 20 kernels per block size, the longest compile of the library -- are spread over three units that
// build side by side: this one (stateSize 14 - 20), pcg_persist_b.hip (22 - 28) and pcg_persist_c.hip (30 - 36).
#include "pcg_persist_kernels.hpp"

namespace gbdpcg {

// ---- host side ---------------------------------------------------------------------------------------------------

// Block sizes with persistent kernels, by the unit that holds them: BASELINE's 14 and 36 and every even size in between (one
// problem of any other size takes the split path, whose hipGraph is 2 max_iter + 4 launches whatever the iteration count).
#define GBDPCG_PERSIST_N_A(X) X(14) X(16) X(18) X(20)
#define GBDPCG_PERSIST_N_B(X) X(22) X(24) X(26) X(28)
#define GBDPCG_PERSIST_N_C(X) X(30) X(32) X(34) X(36)
#define GBDPCG_PERSIST_N(X) GBDPCG_PERSIST_N_A(X) GBDPCG_PERSIST_N_B(X) GBDPCG_PERSIST_N_C(X)

template <typename T> static bool persist_has_kernel(uint32_t n)
{
#define GBDPCG_CASE(NN) \
    if (n == NN) return true;
    GBDPCG_PERSIST_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    return false;
}

// Knots per workgroup for this launch (0: the shape cannot run persistently): every workgroup must be resident at
// once, one per CU.
template <typename T> uint32_t persist_knots_per_wg(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, bool one_reduction)
{
    static const bool off = getenv("GBDPCG_NO_PERSIST") != nullptr;   // tuning runs only
    if (off || !persist_has_kernel<T>(n)) return 0;
    static const int forced = [] {
        const char *e = getenv("GBDPCG_PERSIST_K");   // tuning runs only
        return e ? atoi(e) : 0;
    }();
    // two knots per workgroup measured fastest on config 4 (5.7 us per iteration; 6.2 with three, 6.9 with one)
    for (uint32_t K : {2u, 3u, 1u}) {
        if (forced && (uint32_t)forced != K) continue;
        if (one_reduction && K < 2) continue;   // its first and last own knots carry one halo block-row each
        if ((uint64_t)((N + K - 1) / K) * batch <= (uint64_t)dev.num_cus && (N + K - 1) / K <= 256) return K;
    }
    return 0;
}

template <typename T> size_t persist_workspace_bytes(uint32_t n, uint32_t N, uint32_t batch)
{
    return persist_words<T>(n, N) * sizeof(u64) * batch;
}

template <typename T> size_t persist_rescue_bytes(uint32_t n, uint32_t N, uint32_t batch)
{
    return rescue_vec_elems<T>(n, N) * sizeof(T) * batch;
}

template <typename T>
hipError_t launch_pcg_persist(const DeviceInfo &dev, const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction)
{
    const uint32_t K = persist_knots_per_wg<T>(dev, a.n, a.N, a.batch, one_reduction);
    if (K == 0 || workspace == nullptr) return hipErrorInvalidValue;
#define GBDPCG_CASE(NN) \
    if (a.n == NN) return launch_persist_n<T, NN>(a, workspace, s, one_reduction, K);
    GBDPCG_PERSIST_N_A(GBDPCG_CASE)
#undef GBDPCG_CASE
#define GBDPCG_CASE(NN) \
    if (a.n == NN) return launch_pcg_persist_b<T>(a, workspace, s, one_reduction, K);
    GBDPCG_PERSIST_N_B(GBDPCG_CASE)
#undef GBDPCG_CASE
#define GBDPCG_CASE(NN) \
    if (a.n == NN) return launch_pcg_persist_c<T>(a, workspace, s, one_reduction, K);
    GBDPCG_PERSIST_N_C(GBDPCG_CASE)
#undef GBDPCG_CASE
    return hipErrorInvalidValue;
}

template uint32_t persist_knots_per_wg<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, bool);
template uint32_t persist_knots_per_wg<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, bool);
template size_t persist_workspace_bytes<float>(uint32_t, uint32_t, uint32_t);
template size_t persist_workspace_bytes<double>(uint32_t, uint32_t, uint32_t);
template size_t persist_rescue_bytes<float>(uint32_t, uint32_t, uint32_t);
template size_t persist_rescue_bytes<double>(uint32_t, uint32_t, uint32_t);
template hipError_t launch_pcg_persist<float>(const DeviceInfo &, const PcgArgs<float> &, void *, hipStream_t, bool);
template hipError_t launch_pcg_persist<double>(const DeviceInfo &, const PcgArgs<double> &, void *, hipStream_t, bool);

}  // namespace gbdpcg
