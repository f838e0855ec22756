// symcheck.hip -- is a block-tridiagonal matrix symmetric in the sense the symmetric streaming path
// needs: L_{k+1} == R_k^T bit for bit, for every knot?  One flag per problem.  (D_k itself need not be symmetric: it is always read in full.)
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

// Plain kernels instead of hipMemsetAsync.  In round 1 a solve graph whose flags were initialised by a memset node
// was seen to take the general kernel for every problem when replayed back to back (flags read 0 / garbage) and the
// nodes were replaced by these kernels.  Round 2 could not reproduce it in any of three forms -- a stand-alone HIP
// program (tools/memset_node_probe.cpp), this library rebuilt with memset nodes (-DGBDPCG_FILL_WITH_MEMSET,
// tools/memset_variant_probe.py) and the pre-fix tree itself -- so neither a runtime defect nor a captured-argument
// bug (hipMemsetAsync takes pointer, value and size by value) is established; DESIGN.md section 3 has the record.  The
// kernels stay: they cost the same, and the default one-launch check needs no initialisation at all.
__global__ void fill_bytes_kernel(uint8_t *p, uint8_t v, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}
__global__ void fill_words_kernel(uint32_t *p, uint32_t v, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}
// -DGBDPCG_FILL_WITH_MEMSET (diagnostic variant only, tools/memset_variant_probe.py): the round-1 form, memset nodes.
hipError_t launch_fill_bytes(uint8_t *p, uint8_t v, size_t count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
#ifdef GBDPCG_FILL_WITH_MEMSET
    return hipMemsetAsync(p, v, count, s);
#else
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, v, count);
    return hipGetLastError();
#endif
}
hipError_t launch_fill_words(uint32_t *p, uint32_t v, size_t count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
#ifdef GBDPCG_FILL_WITH_MEMSET
    return v == 0 ? hipMemsetAsync(p, 0, count * sizeof(uint32_t), s) : hipMemsetD32Async((hipDeviceptr_t)p, (int)v, count, s);
#else
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, v, count);
    return hipGetLastError();
#endif
}

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

// Fast form for even n and 16-byte aligned matrices: R_k and L_{k+1} are ADJACENT in memory
// (k*3n^2 + 2n^2 .. (k+1)*3n^2 + n^2), so a pair is one run of 2n^2 elements.  A workgroup reads `ppw`
// pairs of up to two matrices (S and Pinv in the same launch) with dense non-temporal 16-byte loads that
// are all in flight together, parks them in LDS and compares R_k(r,c) with L_{k+1}(c,r) there.
constexpr uint32_t kPairVPT = 4, kPairThreads = 256;

template <typename T>
__global__ __launch_bounds__(kPairThreads) void check_symmetric_pair_kernel(uint32_t n, uint32_t N, uint32_t ppw,
                                                                            uint32_t chunks, const T *__restrict__ A,
                                                                            const T *__restrict__ B,
                                                                            uint8_t *__restrict__ flags, uint32_t per_chunk)
{
    using V4 = typename NtVec<float, 4>::type;  // 16 bytes, whatever T is
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t nn = n * n, pair = 2 * nn;
    const uint32_t prob = blockIdx.x / chunks, chunk = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk * ppw;
    const uint32_t pairs = min(ppw, N - 1 - k0);
    const uint32_t vecs_per_pair = pair * sizeof(T) / 16;
    const uint32_t total_v = pairs * vecs_per_pair;
    const size_t mstride = (size_t)3 * nn * N;
    V4 *bufA = reinterpret_cast<V4 *>(smem_raw), *bufB = bufA + ppw * vecs_per_pair;
    V4 va[kPairVPT], vb[kPairVPT];
#pragma unroll
    for (uint32_t q = 0; q < kPairVPT; ++q) {
        const uint32_t i = threadIdx.x + q * kPairThreads;
        const uint32_t ii = i < total_v ? i : 0u;
        const uint32_t j = ii / vecs_per_pair, v = ii - j * vecs_per_pair;
        const size_t off = (size_t)prob * mstride + (size_t)(k0 + j) * 3 * nn + 2 * nn;  // R_{k0+j}, then L_{k0+j+1}
        va[q] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(A + off) + v);
        if (B) vb[q] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(B + off) + v);
    }
#pragma unroll
    for (uint32_t q = 0; q < kPairVPT; ++q) {
        const uint32_t i = threadIdx.x + q * kPairThreads;
        if (i < total_v) {
            bufA[i] = va[q];
            if (B) bufB[i] = vb[q];
        }
    }
    __syncthreads();
    const T *ea = reinterpret_cast<const T *>(bufA), *eb = reinterpret_cast<const T *>(bufB);
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < pairs * nn; i += kPairThreads) {
        const uint32_t j = i / nn, e = i - j * nn;
        const uint32_t c = e / n, r = e - c * n;
        const uint32_t ri = j * pair + e, li = j * pair + nn + r * n + c;  // R_k(r,c) vs L_{k+1}(c,r)
        bad |= bits_of(ea[ri]) != bits_of(ea[li]);
        if (B) bad |= bits_of(eb[ri]) != bits_of(eb[li]);
    }
    if (per_chunk) {
        // one verdict byte per workgroup, written unconditionally: nothing has to be initialised beforehand
        // (the solve launches AND the bytes of a problem, internal.hpp: pcg_takes)
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) flags[(size_t)prob * chunks + chunk] = any_bad ? 0 : 1;
    } else if (bad) {
        flags[prob] = 0;
    }
}

// Both matrices of a solve in one pass (B may be null).  Returns false when the shape / alignment does
// not fit the pair kernel; the caller then uses launch_check_symmetric per matrix.
template <typename T>
uint32_t check_pair_chunks(uint32_t n, uint32_t N)
{
    if (n % 2 || N < 2) return 0;
    const uint32_t vecs_per_pair = 2 * n * n * sizeof(T) / 16;
    const uint32_t ppw = kPairVPT * kPairThreads / vecs_per_pair;
    return ppw ? (N - 1 + ppw - 1) / ppw : 0;
}

template <typename T>
bool launch_check_symmetric_pair(uint32_t n, uint32_t N, uint32_t batch, const T *A, const T *B, uint8_t *flags,
                                 hipStream_t s, hipError_t *err, uint32_t *verdicts_per_problem)
{
    const uint32_t nn = n * n;
    if (n % 2 || N < 2 || (reinterpret_cast<uintptr_t>(A) % 16) || (B && reinterpret_cast<uintptr_t>(B) % 16)) return false;
    const uint32_t vecs_per_pair = 2 * nn * sizeof(T) / 16;
    const uint32_t ppw = kPairVPT * kPairThreads / vecs_per_pair;
    if (ppw == 0) return false;
    const uint32_t chunks = (N - 1 + ppw - 1) / ppw;
    if (verdicts_per_problem) {
        *verdicts_per_problem = chunks;  // flags holds batch * chunks bytes, every one written by the kernel
    } else {
        *err = launch_fill_bytes(flags, 1, batch, s);
        if (*err != hipSuccess) return true;
    }
    const size_t lds = (size_t)2 * ppw * vecs_per_pair * 16;
    hipLaunchKernelGGL(check_symmetric_pair_kernel<T>, dim3(batch * chunks), dim3(kPairThreads), lds, s, n, N, ppw, chunks,
                       A, B, flags, verdicts_per_problem ? 1u : 0u);
    *err = hipGetLastError();
    return true;
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
        hipError_t e = launch_fill_bytes(flags, 1, batch, s);
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

template bool launch_check_symmetric_pair<float>(uint32_t, uint32_t, uint32_t, const float *, const float *, uint8_t *,
                                                 hipStream_t, hipError_t *, uint32_t *);
template bool launch_check_symmetric_pair<double>(uint32_t, uint32_t, uint32_t, const double *, const double *, uint8_t *,
                                                  hipStream_t, hipError_t *, uint32_t *);
template uint32_t check_pair_chunks<float>(uint32_t, uint32_t);
template uint32_t check_pair_chunks<double>(uint32_t, uint32_t);
template hipError_t launch_check_symmetric<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *, uint8_t *,
                                                  bool, hipStream_t);
template hipError_t launch_check_symmetric<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                                   uint8_t *, bool, hipStream_t);

}  // namespace gbdpcg
