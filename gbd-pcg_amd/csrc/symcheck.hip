// symcheck.hip -- is a block-tridiagonal matrix symmetric in the sense the symmetric streaming path
// needs: L_{k+1} == R_k^T bit for bit, for every knot?  One flag per problem.  (D_k itself need not be symmetric: it is always read in full.)
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

__device__ __forceinline__ uint32_t bits_of(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint64_t bits_of(double v) { return __builtin_bit_cast(uint64_t, v); }

// One workgroup tests KPW consecutive knot pairs (R_k, L_{k+1}) of one problem: both blocks are read
// with dense, coalesced loads; L is parked in LDS so that the transposed comparison happens on chip.
// A workgroup that finds a difference stores 0 into the problem's flag (flags start at 1; every
// writer writes the same value, so the race is benign).
constexpr uint32_t kSymKPW = 8;

template <typename T>
__global__ __launch_bounds__(256) void check_symmetric_kernel(uint32_t n, uint32_t N, uint32_t chunks,
                                                              const T *__restrict__ M, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *tile = reinterpret_cast<T *>(smem_raw);  // [KPW][n*n] : L_{k+1}, column-major
    const uint32_t nn = n * n;
    const uint32_t prob = blockIdx.x / chunks, chunk = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk * kSymKPW;
    const uint32_t pairs = min(kSymKPW, N - 1 - k0);
    const T *Mp = M + (size_t)prob * 3 * nn * N;
    const uint32_t total = pairs * nn;
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint32_t j = i / nn, e = i - j * nn;
        tile[i] = Mp[(size_t)(k0 + j + 1) * 3 * nn + e];                 // L_{k+1}, element e
    }
    __syncthreads();
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint32_t j = i / nn, e = i - j * nn;
        const uint32_t c = e / n, r = e - c * n;
        const T right = Mp[(size_t)(k0 + j) * 3 * nn + 2 * nn + e];      // R_k(r, c)
        const T left = tile[j * nn + r * n + c];                          // L_{k+1}(c, r)
        bad |= bits_of(right) != bits_of(left);
    }
    if (bad) flags[prob] = 0;
}

template <typename T>
hipError_t launch_check_symmetric(const DeviceInfo &, uint32_t n, uint32_t N, uint32_t batch, const T *M, uint8_t *flags,
                                  bool and_into, hipStream_t s)
{
    if (!and_into) {
        hipError_t e = hipMemsetAsync(flags, 1, batch, s);
        if (e != hipSuccess) return e;
    }
    if (N < 2) return hipSuccess;  // a single knot has no off-diagonal blocks
    const uint32_t chunks = (N - 1 + kSymKPW - 1) / kSymKPW;
    const size_t lds = (size_t)kSymKPW * n * n * sizeof(T);
    auto kern = check_symmetric_kernel<T>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(batch * chunks), dim3(256), lds, s, n, N, chunks, M, flags);
    return hipGetLastError();
}

template hipError_t launch_check_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *, uint8_t *,
                                                  bool, hipStream_t);
template hipError_t launch_check_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                                   uint8_t *, bool, hipStream_t);

}  // namespace gbdpcg
