// symcheck.hip -- is a block-tridiagonal matrix symmetric in the sense the symmetric streaming path
// needs: L_{k+1} == R_k^T bit for bit, for every knot?  One flag per problem.  (D_k itself need not be symmetric: it is always read in full.)
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

__device__ __forceinline__ uint32_t bits_of(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint64_t bits_of(double v) { return __builtin_bit_cast(uint64_t, v); }

// One workgroup tests `kpw` consecutive knot pairs (R_k, L_{k+1}) of one problem (kpw * n^2 <= 2048
// elements, at most 8 per thread): both blocks are read with dense, coalesced loads that are all in
// flight before the first use; L is parked in LDS so that the transposed comparison happens on chip.
// A workgroup that finds a difference stores 0 into the problem's flag (flags start at 1; every
// writer writes the same value, so the race is benign).
constexpr uint32_t kSymEPT = 8, kSymThreads = 256;

template <typename T>
__global__ __launch_bounds__(kSymThreads) void check_symmetric_kernel(uint32_t n, uint32_t N, uint32_t kpw, uint32_t chunks,
                                                                      const T *__restrict__ M, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *tile = reinterpret_cast<T *>(smem_raw);  // [kpw][n*n] : L_{k+1}, column-major
    const uint32_t nn = n * n;
    const uint32_t prob = blockIdx.x / chunks, chunk = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk * kpw;
    const uint32_t pairs = min(kpw, N - 1 - k0);
    const T *Mp = M + (size_t)prob * 3 * nn * N;
    const uint32_t total = pairs * nn;
    T left[kSymEPT], right[kSymEPT];
#pragma unroll
    for (uint32_t q = 0; q < kSymEPT; ++q) {
        const uint32_t i = threadIdx.x + q * kSymThreads;
        const uint32_t ii = i < total ? i : 0u;
        const uint32_t j = ii / nn, e = ii - j * nn;
        left[q] = Mp[(size_t)(k0 + j + 1) * 3 * nn + e];                 // L_{k+1}, element e
        right[q] = Mp[(size_t)(k0 + j) * 3 * nn + 2 * nn + e];           // R_k,     element e = (r, c)
    }
#pragma unroll
    for (uint32_t q = 0; q < kSymEPT; ++q) {
        const uint32_t i = threadIdx.x + q * kSymThreads;
        if (i < total) tile[i] = left[q];
    }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (uint32_t q = 0; q < kSymEPT; ++q) {
        const uint32_t i = threadIdx.x + q * kSymThreads;
        if (i < total) {
            const uint32_t j = i / nn, e = i - j * nn;
            const uint32_t c = e / n, r = e - c * n;
            bad |= bits_of(right[q]) != bits_of(tile[j * nn + r * n + c]);  // R_k(r,c) vs L_{k+1}(c,r)
        }
    }
    if (bad) flags[prob] = 0;
}

// Fallback for blocks too large for the register-staged kernel (n^2 > 2048): plain strided compare.
template <typename T>
__global__ __launch_bounds__(256) void check_symmetric_big_kernel(uint32_t n, uint32_t N, const T *__restrict__ M,
                                                                  uint8_t *__restrict__ flags)
{
    const size_t nn = (size_t)n * n;
    const uint32_t prob = blockIdx.x / (N - 1), k = blockIdx.x - prob * (N - 1);
    const T *Mp = M + (size_t)prob * 3 * nn * N;
    bool bad = false;
    for (size_t e = threadIdx.x; e < nn; e += 256) {
        const uint32_t c = (uint32_t)(e / n), r = (uint32_t)(e - (size_t)c * n);
        bad |= bits_of(Mp[(size_t)k * 3 * nn + 2 * nn + e]) != bits_of(Mp[(size_t)(k + 1) * 3 * nn + (size_t)r * n + c]);
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
    const uint32_t nn = n * n, cap = kSymEPT * kSymThreads;
    if (nn > cap) {
        hipLaunchKernelGGL(check_symmetric_big_kernel<T>, dim3(batch * (N - 1)), dim3(256), 0, s, n, N, M, flags);
        return hipGetLastError();
    }
    const uint32_t kpw = cap / nn;
    const uint32_t chunks = (N - 1 + kpw - 1) / kpw;
    const size_t lds = (size_t)kpw * nn * sizeof(T);
    hipLaunchKernelGGL(check_symmetric_kernel<T>, dim3(batch * chunks), dim3(kSymThreads), lds, s, n, N, kpw, chunks, M, flags);
    return hipGetLastError();
}

template hipError_t launch_check_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *, uint8_t *,
                                                  bool, hipStream_t);
template hipError_t launch_check_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                                   uint8_t *, bool, hipStream_t);

}  // namespace gbdpcg
