// schur.hip -- the two MPCGPU steps either side of the PCG solve:
//   form_schur     : KKT blocks (Q_k, R_k, A_k, B_k, q_k, r_k, c_k) of a batch of linearised MPC problems
//                    -> S = C G^-1 C' in the [L | D | R] layout pcg<T> reads, gamma = -(c + C G^-1 g), and G^-1
//   recover_primal : lambda -> z = -G^-1 (g + C' lambda)
//
// The reference tree has no code for either (/root/reference/README.md:2-11 states only the system
// Pinv S lambda = Pinv gamma that comes out of the first, README.md:66-77 cites the paper that describes them; MPCGPU
// builds them out of tree with the block helpers of include/utils.cuh:96-161), so there is nothing to be identical to:
// the convention is written out in include/gbdpcg.h and oracle/schur_oracle.py, and the tests hold the results against
// a dense fp64 solve of the whole KKT system.
//
// Block formulas (j = k-1):
//     D_0 = Q_0^-1                                               gamma_0 = -(c_0 + Q_0^-1 q_0)
//     D_k = A_j Q_j^-1 A_j' + B_j R_j^-1 B_j' + Q_k^-1           gamma_k = -(c_k + Q_k^-1 q_k - A_j Q_j^-1 q_j - B_j R_j^-1 r_j)
//     L_k = -A_j Q_j^-1          R_k = -Q_k^-1 A_k'
//
// Work split: ONE WAVEFRONT per block-row (problem, k).  It inverts Q_k, Q_{k-1}, R_{k-1} itself (Gauss-Jordan without
// pivoting on an LDS tableau: the cost blocks are positive definite) -- the wave of row k+1 inverts Q_k again rather than
// wait for this one: the arithmetic is free next to the 4.7 KB a row moves, and no launch boundary, atomics or ordering
// between waves is needed.  Every inverse is mirrored across its diagonal before use; R_k (row k) and L_{k+1} (row k+1)
// are then the same fma chains over the same numbers, so S comes out EXACTLY symmetric in storage (L_{k+1} == R_k'
// bit for bit) and the solve takes its symmetric-storage kernels (gbdpcg_set_symmetric, mode 2 test passes).
// LDS operations of one wave execute in program order, so the synchronisation inside a wave is a compiler fence
// (group_sync<64> of pinv.hip restated).  A row's working set is 7 nx^2 + 3 nu^2 + ... elements of LDS (7.5 KB at
// nx = 14, nu = 7, fp32): four waves per workgroup while they fit 64 KB, one otherwise.
#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

namespace {

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct KktDims {
    uint32_t nx, nu, N;
    uint32_t sg, sc, sv;  // strides of one knot in G / C / g
    size_t szG, szC, szg, szc;
    __host__ __device__ KktDims(uint32_t nx_, uint32_t nu_, uint32_t N_) : nx(nx_), nu(nu_), N(N_)
    {
        sg = nx * nx + nu * nu;
        sc = nx * nx + nx * nu;
        sv = nx + nu;
        szG = (size_t)sg * N - nu * nu;
        szC = (size_t)sc * (N - 1);
        szg = (size_t)sv * N - nu;
        szc = (size_t)nx * N;
    }
};

// LDS elements one wave of the formation kernel needs (kept a multiple of 4 so that every wave's block starts 16-byte aligned).
__host__ __device__ inline uint32_t schur_wave_elems(uint32_t nx, uint32_t nu)
{
    const uint32_t m = nx > nu ? nx : nu;
    const uint32_t e = 2 * m * m + 3 * m      // tableau, scaled pivot row, pivot column
                       + 5 * nx * nx          // Qc, Qp, Ap, Ac, W
                       + nu * nu              // Rp
                       + 2 * nx * nu          // Bp, V
                       + 3 * nx + nu;         // q_k, q_j, c_k, r_j
    return (e + 3u) & ~3u;
}
__host__ __device__ inline uint32_t recover_wave_elems(uint32_t nx, uint32_t nu)
{
    const uint32_t e = 2 * nx * nx + nu * nu + nx * nu + 3 * nx + nu;  // Qi, A, Ri, B, lambda_{k+1}, t_x, (spare), t_u
    return (e + 3u) & ~3u;
}

// Inverse of the m x m block at `src` (global, column-major) into `out` (LDS, column-major, mirrored across the diagonal).
// Same arithmetic, element for element, as pinv_diag_kernel (pinv.hip): pr = row_j * (1/pivot), a_rc = fma(-a_rj, pr_c, a_rc).
template <typename T>
__device__ __forceinline__ void wave_invert(const T *__restrict__ src, uint32_t m, T *tab, T *prow, T *pcol, T *out, uint32_t lane)
{
    const uint32_t w = 2 * m;
    for (uint32_t i = lane; i < m * m; i += 64) {
        const uint32_t c = i / m, r = i - c * m;
        tab[r * w + c] = src[i];
        tab[r * w + m + c] = (r == c) ? T(1) : T(0);
    }
    wave_sync();
    for (uint32_t j = 0; j < m; ++j) {
        const T piv = T(1) / tab[j * w + j];
        for (uint32_t c = lane; c < w; c += 64) prow[c] = tab[j * w + c] * piv;
        for (uint32_t r = lane; r < m; r += 64) pcol[r] = tab[r * w + j];
        wave_sync();
        uint32_t r = lane / w, c = lane - r * w;  // element lane, lane + 64, ... of the tableau without a division per element
        const uint32_t dr = 64 / w, dc = 64 - dr * w;
        for (uint32_t i = lane; i < m * w; i += 64) {
            tab[i] = (r == j) ? prow[c] : fma_t(-pcol[r], prow[c], tab[i]);
            r += dr;
            c += dc;
            if (c >= w) {
                c -= w;
                ++r;
            }
        }
        wave_sync();
    }
    for (uint32_t i = lane; i < m * m; i += 64) {
        const uint32_t c = i / m, r = i - c * m;
        out[i] = r <= c ? tab[r * w + m + c] : tab[c * w + m + r];
    }
    wave_sync();
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void schur_form_kernel(uint32_t nx, uint32_t nu, uint32_t N, uint64_t rows, const T *__restrict__ G,
                                                        const T *__restrict__ C, const T *__restrict__ g, const T *__restrict__ c,
                                                        T *__restrict__ S, T *__restrict__ gamma, T *__restrict__ Ginv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t row = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows) return;  // whole wave
    const KktDims d(nx, nu, N);
    const uint64_t prob = row / N;
    const uint32_t k = (uint32_t)(row - prob * N);
    const uint32_t nn = nx * nx, uu = nu * nu, xu = nx * nu, m = nx > nu ? nx : nu;

    T *tab = reinterpret_cast<T *>(smem_raw) + (size_t)wave * schur_wave_elems(nx, nu);
    T *prow = tab + 2 * m * m, *pcol = prow + 2 * m;
    T *Qc = pcol + m, *Qp = Qc + nn, *Ap = Qp + nn, *Ac = Ap + nn, *W = Ac + nn;
    T *Rp = W + nn, *Bp = Rp + uu, *V = Bp + xu;
    T *qc = V + xu, *qp = qc + nx, *ck = qp + nx, *rp = ck + nx;

    const T *Gp = G + prob * d.szG, *Cp = C + prob * d.szC, *gp = g + prob * d.szg, *cp = c + prob * d.szc;
    T *Sk = S + (size_t)row * 3 * nn;
    T *Gi = Ginv ? Ginv + prob * d.szG : nullptr;

    // this knot: Q_k^-1, q_k, c_k, A_k
    wave_invert(Gp + (size_t)k * d.sg, nx, tab, prow, pcol, Qc, lane);
    if (Gi)
        for (uint32_t i = lane; i < nn; i += 64) Gi[(size_t)k * d.sg + i] = Qc[i];
    for (uint32_t i = lane; i < nx; i += 64) {
        qc[i] = gp[(size_t)k * d.sv + i];
        ck[i] = cp[(size_t)k * nx + i];
    }
    const bool has_next = k + 1 < N, has_prev = k > 0;
    if (has_next)
        for (uint32_t i = lane; i < nn; i += 64) Ac[i] = Cp[(size_t)k * d.sc + i];
    if (has_prev) {
        const uint32_t j = k - 1;
        for (uint32_t i = lane; i < nn; i += 64) Ap[i] = Cp[(size_t)j * d.sc + i];
        for (uint32_t i = lane; i < xu; i += 64) Bp[i] = Cp[(size_t)j * d.sc + nn + i];
        for (uint32_t i = lane; i < nx; i += 64) qp[i] = gp[(size_t)j * d.sv + i];
        for (uint32_t i = lane; i < nu; i += 64) rp[i] = gp[(size_t)j * d.sv + nx + i];
        wave_invert(Gp + (size_t)j * d.sg, nx, tab, prow, pcol, Qp, lane);
        wave_invert(Gp + (size_t)j * d.sg + nn, nu, tab, prow, pcol, Rp, lane);
        if (Gi)
            for (uint32_t i = lane; i < uu; i += 64) Gi[(size_t)j * d.sg + nn + i] = Rp[i];
        // W = A_j Q_j^-1 (L_k = -W),  V = B_j R_j^-1
        for (uint32_t i = lane; i < nn; i += 64) {
            const uint32_t cc = i / nx, r = i - cc * nx;
            T acc = T(0);
            for (uint32_t q = 0; q < nx; ++q) acc = fma_t(Ap[q * nx + r], Qp[cc * nx + q], acc);
            W[i] = acc;
            Sk[i] = -acc;
        }
        for (uint32_t i = lane; i < xu; i += 64) {
            const uint32_t cc = i / nx, r = i - cc * nx;
            T acc = T(0);
            for (uint32_t q = 0; q < nu; ++q) acc = fma_t(Bp[q * nx + r], Rp[cc * nu + q], acc);
            V[i] = acc;
        }
    } else {
        for (uint32_t i = lane; i < nn; i += 64) Sk[i] = T(0);
    }
    wave_sync();
    // D_k = W A_j' + V B_j' + Q_k^-1
    for (uint32_t i = lane; i < nn; i += 64) {
        const uint32_t cc = i / nx, r = i - cc * nx;
        T acc = T(0);
        if (has_prev) {
            for (uint32_t q = 0; q < nx; ++q) acc = fma_t(W[q * nx + r], Ap[q * nx + cc], acc);
            for (uint32_t q = 0; q < nu; ++q) acc = fma_t(V[q * nx + r], Bp[q * nx + cc], acc);
        }
        Sk[nn + i] = acc + Qc[i];
    }
    // R_k = -Q_k^-1 A_k': element (r, cc) = -sum_q Qinv(r, q) A(cc, q) -- the chain L_{k+1}(cc, r) runs over the mirrored inverse
    for (uint32_t i = lane; i < nn; i += 64) {
        const uint32_t cc = i / nx, r = i - cc * nx;
        T acc = T(0);
        if (has_next)
            for (uint32_t q = 0; q < nx; ++q) acc = fma_t(Ac[q * nx + cc], Qc[r * nx + q], acc);
        Sk[2 * nn + i] = has_next ? -acc : T(0);
    }
    // gamma_k
    for (uint32_t r = lane; r < nx; r += 64) {
        T v = ck[r];
        for (uint32_t q = 0; q < nx; ++q) v = fma_t(Qc[q * nx + r], qc[q], v);
        if (has_prev) {
            T s = T(0);
            for (uint32_t q = 0; q < nx; ++q) s = fma_t(W[q * nx + r], qp[q], s);
            for (uint32_t q = 0; q < nu; ++q) s = fma_t(V[q * nx + r], rp[q], s);
            v -= s;
        }
        gamma[(size_t)row * nx + r] = -v;
    }
}

// z = -G^-1 (g + C' lambda): x_k = -Q_k^-1 (q_k + lambda_k - A_k' lambda_{k+1}),  u_k = -R_k^-1 (r_k - B_k' lambda_{k+1}).
template <typename T>
__global__ __launch_bounds__(256) void schur_recover_kernel(uint32_t nx, uint32_t nu, uint32_t N, uint64_t rows,
                                                           const T *__restrict__ Ginv, const T *__restrict__ C,
                                                           const T *__restrict__ g, const T *__restrict__ lambda, T *__restrict__ z)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t row = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows) return;
    const KktDims d(nx, nu, N);
    const uint64_t prob = row / N;
    const uint32_t k = (uint32_t)(row - prob * N);
    const uint32_t nn = nx * nx, uu = nu * nu, xu = nx * nu;
    const bool has_next = k + 1 < N;

    T *Qi = reinterpret_cast<T *>(smem_raw) + (size_t)wave * recover_wave_elems(nx, nu);
    T *A = Qi + nn, *Ri = A + nn, *B = Ri + uu, *ln = B + xu, *tx = ln + nx, *tu = tx + 2 * nx;
    const T *Gi = Ginv + prob * d.szG + (size_t)k * d.sg, *Ck = C + prob * d.szC + (size_t)k * d.sc;
    const T *gk = g + prob * d.szg + (size_t)k * d.sv;
    T *zk = z + prob * d.szg + (size_t)k * d.sv;

    for (uint32_t i = lane; i < nn; i += 64) Qi[i] = Gi[i];
    if (has_next) {
        for (uint32_t i = lane; i < nn; i += 64) A[i] = Ck[i];
        for (uint32_t i = lane; i < uu; i += 64) Ri[i] = Gi[nn + i];
        for (uint32_t i = lane; i < xu; i += 64) B[i] = Ck[nn + i];
        for (uint32_t i = lane; i < nx; i += 64) ln[i] = lambda[(size_t)(row + 1) * nx + i];
    }
    wave_sync();
    for (uint32_t r = lane; r < nx; r += 64) {
        T t = gk[r] + lambda[(size_t)row * nx + r];
        if (has_next) {
            T s = T(0);
            for (uint32_t q = 0; q < nx; ++q) s = fma_t(A[r * nx + q], ln[q], s);  // (A' lambda)_r = sum_q A(q, r) lambda_q
            t -= s;
        }
        tx[r] = t;
    }
    if (has_next)
        for (uint32_t r = lane; r < nu; r += 64) {
            T s = T(0);
            for (uint32_t q = 0; q < nx; ++q) s = fma_t(B[r * nx + q], ln[q], s);
            tu[r] = gk[nx + r] - s;
        }
    wave_sync();
    for (uint32_t r = lane; r < nx; r += 64) {
        T s = T(0);
        for (uint32_t q = 0; q < nx; ++q) s = fma_t(Qi[q * nx + r], tx[q], s);
        zk[r] = -s;
    }
    if (has_next)
        for (uint32_t r = lane; r < nu; r += 64) {
            T s = T(0);
            for (uint32_t q = 0; q < nu; ++q) s = fma_t(Ri[q * nu + r], tu[q], s);
            zk[nx + r] = -s;
        }
}

// Waves per workgroup for a per-wave LDS need; 0 = does not fit one CU.
static uint32_t waves_for(const DeviceInfo &dev, size_t wave_bytes)
{
    if (wave_bytes > dev.lds_per_wg_max) return 0;
    uint32_t w = 4;
    while (w > 1 && w * wave_bytes > 64 * 1024) --w;
    return w;
}

template <typename T>
hipError_t launch_form_schur(const DeviceInfo &dev, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *G, const T *C,
                             const T *g, const T *c, T *S, T *gamma, T *Ginv, hipStream_t s)
{
    const size_t wave_bytes = (size_t)schur_wave_elems(nx, nu) * sizeof(T);
    const uint32_t waves = waves_for(dev, wave_bytes);
    if (!waves) return hipErrorInvalidValue;
    const uint64_t rows = (uint64_t)batch * N;
    const uint64_t grid = (rows + waves - 1) / waves;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = waves * wave_bytes;
    auto kern = schur_form_kernel<T>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(64 * waves), lds, s, nx, nu, N, rows, G, C, g, c, S, gamma, Ginv);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_recover_primal(const DeviceInfo &dev, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *Ginv,
                                 const T *C, const T *g, const T *lambda, T *z, hipStream_t s)
{
    const size_t wave_bytes = (size_t)recover_wave_elems(nx, nu) * sizeof(T);
    const uint32_t waves = waves_for(dev, wave_bytes);
    if (!waves) return hipErrorInvalidValue;
    const uint64_t rows = (uint64_t)batch * N;
    const uint64_t grid = (rows + waves - 1) / waves;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = waves * wave_bytes;
    auto kern = schur_recover_kernel<T>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(64 * waves), lds, s, nx, nu, N, rows, Ginv, C, g, lambda, z);
    return hipGetLastError();
}

template <typename T> bool schur_shape_ok(const DeviceInfo &dev, uint32_t nx, uint32_t nu)
{
    return nx >= 1 && nu >= 1 && waves_for(dev, (size_t)schur_wave_elems(nx, nu) * sizeof(T)) != 0;
}

template hipError_t launch_form_schur<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, uint32_t, const float *, const float *,
                                             const float *, const float *, float *, float *, float *, hipStream_t);
template hipError_t launch_form_schur<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, uint32_t, const double *, const double *,
                                              const double *, const double *, double *, double *, double *, hipStream_t);
template hipError_t launch_recover_primal<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, uint32_t, const float *,
                                                 const float *, const float *, const float *, float *, hipStream_t);
template hipError_t launch_recover_primal<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, uint32_t, const double *,
                                                  const double *, const double *, const double *, double *, hipStream_t);
template bool schur_shape_ok<float>(const DeviceInfo &, uint32_t, uint32_t);
template bool schur_shape_ok<double>(const DeviceInfo &, uint32_t, uint32_t);

}  // namespace gbdpcg
