// symcheck.hip -- is a block-tridiagonal matrix symmetric in the sense the symmetric streaming path
// needs: L_{k+1} == R_k^T bit for bit, for every knot?  One workgroup per problem, one flag per
// problem.  (D_k itself need not be symmetric: it is always read in full.)
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

__device__ __forceinline__ uint32_t bits_of(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint64_t bits_of(double v) { return __builtin_bit_cast(uint64_t, v); }

template <typename T>
__global__ __launch_bounds__(256) void check_symmetric_kernel(uint32_t n, uint32_t N, const T *__restrict__ M,
                                                              uint8_t *__restrict__ flags)
{
    __shared__ int bad;
    const size_t nn = (size_t)n * n;
    const T *Mp = M + (size_t)blockIdx.x * 3 * nn * N;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    const size_t total = (size_t)(N - 1) * nn;
    for (size_t i = threadIdx.x; i < total && !mine; i += 256) {
        const size_t k = i / nn, e = i - k * nn;
        const uint32_t c = (uint32_t)(e / n), r = (uint32_t)(e - (size_t)c * n);
        const T right = Mp[k * 3 * nn + 2 * nn + (size_t)c * n + r];        // R_k(r, c)
        const T left = Mp[(k + 1) * 3 * nn + (size_t)r * n + c];            // L_{k+1}(c, r)
        // bitwise: -0.0 vs 0.0 or NaNs would change nothing in the product but are not "the same numbers"
        if (bits_of(right) != bits_of(left)) mine = 1;
    }
    if (mine) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = bad ? 0 : 1;
}

template <typename T>
hipError_t launch_check_symmetric(const DeviceInfo &, uint32_t n, uint32_t N, uint32_t batch, const T *M, uint8_t *flags,
                                  hipStream_t s)
{
    hipLaunchKernelGGL(check_symmetric_kernel<T>, dim3(batch), dim3(256), 0, s, n, N, M, flags);
    return hipGetLastError();
}

template hipError_t launch_check_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *, uint8_t *,
                                                  hipStream_t);
template hipError_t launch_check_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                                   uint8_t *, hipStream_t);

}  // namespace gbdpcg
