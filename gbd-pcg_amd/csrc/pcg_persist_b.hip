// pcg_persist_b.hip -- the persistent kernels (pcg_persist_kernels.hpp) of one group of block sizes; pcg_persist.hip dispatches here.
#include "pcg_persist_kernels.hpp"

namespace gbdpcg {

template <typename T>
hipError_t launch_pcg_persist_b(const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction, uint32_t K)
{
#define GBDPCG_CASE(NN) \
    if (a.n == NN) return launch_persist_n<T, NN>(a, workspace, s, one_reduction, K);
#define GBDPCG_SIZES(X) X(22) X(24) X(26) X(28)
    GBDPCG_SIZES(GBDPCG_CASE)
#undef GBDPCG_SIZES
#undef GBDPCG_CASE
    return hipErrorInvalidValue;
}

template hipError_t launch_pcg_persist_b<float>(const PcgArgs<float> &, void *, hipStream_t, bool, uint32_t);
template hipError_t launch_pcg_persist_b<double>(const PcgArgs<double> &, void *, hipStream_t, bool, uint32_t);

}  // namespace gbdpcg
