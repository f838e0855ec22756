// pinv.hip -- build the preconditioner Pinv from S on the device.
//
// The reference never forms Pinv itself: its host overload leaves d_Pinv uninitialised
// (/root/reference/include/interface.cuh:45-46,57-59) and MPCGPU builds it out of tree with the
// block helpers load_block_bd / store_block_bd (include/utils.cuh:96-161).  This file supplies the
// step so the host-pointer API is usable end to end:
//   IDENTITY     : D slot = I, L/R slots = 0
//   BLOCK_JACOBI : D slot = D_k^-1
//   STAIR        : D slot = D_k^-1, L slot = -D_k^-1 L_k D_{k-1}^-1, R slot = -D_k^-1 R_k D_{k+1}^-1
// (the symmetric-stair preconditioner of the MPCGPU paper the README cites, README.md:66-77).
//
// One workgroup per (problem, knot).  Pass 1 inverts D_k by Gauss-Jordan on an LDS-resident
// [D | I] tableau (no pivoting: D_k is a definite diagonal block of a Schur complement); pass 2
// (STAIR only) does the two triple products with all three operands in LDS.  This is O(n^3)
// per knot on n^2 data and runs once per control step; it is not the bandwidth-bound hot loop.
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

constexpr int kPinvThreads = 256;

template <typename T>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_kernel(uint32_t n, uint32_t N, const T *__restrict__ S,
                                                                T *__restrict__ Pinv, int kind)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *tab = reinterpret_cast<T *>(smem_raw);  // [n][2n] row-major tableau
    T *colj = tab + 2 * n * n;                 // [n] column j before elimination
    const uint32_t tid = threadIdx.x;
    const size_t blk = (size_t)blockIdx.x * 3 * n * n;  // (problem, knot) flattened: same stride
    const T *D = S + blk + (size_t)n * n;
    T *out = Pinv + blk;
    const uint32_t w = 2 * n;

    for (uint32_t i = tid; i < n * n; i += kPinvThreads) {
        const uint32_t c = i / n, r = i - c * n;  // column-major source
        tab[r * w + c] = (kind == 0) ? (r == c ? T(1) : T(0)) : D[i];
        tab[r * w + n + c] = (r == c) ? T(1) : T(0);
    }
    __syncthreads();
    if (kind != 0) {
        for (uint32_t j = 0; j < n; ++j) {
            const T piv = T(1) / tab[j * w + j];
            for (uint32_t r = tid; r < n; r += kPinvThreads) colj[r] = tab[r * w + j];
            __syncthreads();
            for (uint32_t i = tid; i < n * w; i += kPinvThreads) {
                const uint32_t r = i / w, c = i - r * w;
                const T pr = tab[j * w + c] * piv;  // scaled pivot-row entry
                if (r == j) continue;
                tab[r * w + c] = fma_t(-colj[r], pr, tab[r * w + c]);
            }
            __syncthreads();
            for (uint32_t c = tid; c < w; c += kPinvThreads) tab[j * w + c] *= piv;
            __syncthreads();
        }
    }
    // D_k^-1 of a symmetric D_k is symmetric in exact arithmetic but not bit for bit after the
    // elimination; the upper triangle is mirrored so that the stair blocks built from it come out
    // exactly symmetric (L_{k+1} == R_k^T) whenever S is -- what gbdpcg_set_symmetric relies on.
    for (uint32_t i = tid; i < n * n; i += kPinvThreads) {
        const uint32_t c = i / n, r = i - c * n;
        out[(size_t)n * n + i] = r <= c ? tab[r * w + n + c] : tab[c * w + n + r];
        out[i] = T(0);
        out[(size_t)2 * n * n + i] = T(0);
    }
}

// Off-diagonal slots of the stair preconditioner; reads the D slots pass 1 wrote.
template <typename T>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_kernel(uint32_t n, uint32_t N, const T *__restrict__ S,
                                                                 T *Pinv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *A = reinterpret_cast<T *>(smem_raw);  // D_k^-1        (column-major)
    T *B = A + n * n;                        // O_k           (L_k or R_k)
    T *C = B + n * n;                        // D_{k+-1}^-1
    T *W = C + n * n;                        // A*B
    const uint32_t tid = threadIdx.x;
    const uint32_t k = blockIdx.x % N;
    const size_t blk = (size_t)blockIdx.x * 3 * n * n;
    const uint32_t nn = n * n;

    // Right slot:  R'_k     = -(D_k^-1 R_k) D_{k+1}^-1.
    // Left slot :  L'_k     = -D_k^-1 L_k D_{k-1}^-1, evaluated as the TRANSPOSE of
    //              X = -(D_{k-1}^-1 L_k^T) D_k^-1  -- the same operation sequence the right slot of
    //              knot k-1 runs on (D_{k-1}^-1, R_{k-1}, D_k^-1).  With symmetric D^-1 blocks the two are
    //              equal as matrices for any S, and bit for bit each other's transpose when L_k == R_{k-1}^T.
    for (int side = 0; side < 2; ++side) {  // 0: left slot (needs k-1), 1: right slot (needs k+1)
        if ((side == 0 && k == 0) || (side == 1 && k == N - 1)) continue;
        const size_t nb = side == 0 ? blk - (size_t)3 * nn : blk + (size_t)3 * nn;
        for (uint32_t i = tid; i < nn; i += kPinvThreads) {
            const uint32_t c = i / n, r = i - c * n;
            if (side == 1) {
                A[i] = Pinv[blk + nn + i];            // D_k^-1
                B[i] = S[blk + 2 * (size_t)nn + i];   // R_k
                C[i] = Pinv[nb + nn + i];             // D_{k+1}^-1
            } else {
                A[i] = Pinv[nb + nn + i];             // D_{k-1}^-1
                B[i] = S[blk + (size_t)r * n + c];    // L_k^T : element (r,c) = L_k(c,r)
                C[i] = Pinv[blk + nn + i];            // D_k^-1
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < nn; i += kPinvThreads) {
            const uint32_t c = i / n, r = i - c * n;
            T acc = T(0);
            for (uint32_t q = 0; q < n; ++q) acc = fma_t(A[q * n + r], B[c * n + q], acc);
            W[i] = acc;
        }
        __syncthreads();
        for (uint32_t i = tid; i < nn; i += kPinvThreads) {
            const uint32_t c = i / n, r = i - c * n;
            T acc = T(0);
            for (uint32_t q = 0; q < n; ++q) acc = fma_t(W[q * n + r], C[c * n + q], acc);
            if (side == 1) Pinv[blk + 2 * (size_t)nn + i] = -acc;                  // R'_k(r,c)
            else Pinv[blk + (size_t)r * n + c] = -acc;                             // L'_k(c,r) = X(r,c)
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_form_pinv(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *S, T *Pinv,
                            int kind, hipStream_t s)
{
    const size_t lds1 = ((size_t)2 * n * n + n) * sizeof(T);
    const size_t lds2 = (size_t)4 * n * n * sizeof(T);
    if (lds1 > dev.lds_per_wg_max || lds2 > dev.lds_per_wg_max) return hipErrorInvalidValue;
    const uint64_t blocks = (uint64_t)N * batch;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    auto k1 = pinv_diag_kernel<T>;
    auto k2 = pinv_stair_kernel<T>;
    if (lds1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    if (lds2 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k1, dim3((uint32_t)blocks), dim3(kPinvThreads), lds1, s, n, N, S, Pinv, kind);
    if (kind == 2) hipLaunchKernelGGL(k2, dim3((uint32_t)blocks), dim3(kPinvThreads), lds2, s, n, N, S, Pinv);
    return hipGetLastError();
}

template hipError_t launch_form_pinv<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *,
                                            float *, int, hipStream_t);
template hipError_t launch_form_pinv<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                             double *, int, hipStream_t);

}  // namespace gbdpcg
