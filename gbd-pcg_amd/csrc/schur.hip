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
#include <cstdlib>

#include "bt_device.hpp"
#include "internal.hpp"

#ifndef GBDPCG_SCHUR_SKIP
#define GBDPCG_SCHUR_SKIP 0   // timing builds only: 1 no elimination, 2 no products, 4 no stores, 8 no requests after the first (results are wrong)
#endif

#ifdef GBDPCG_SCHUR_STAMPS   // timing builds only: s_memtime at the phase boundaries of one step of workgroup 0, left BEHIND gamma (tools/schur_run.py --stamps allocates 96 bytes more)
#define SCHUR_STAMP(i) do { if (stamp_now) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st[i] = t_; } } while (0)
#else
#define SCHUR_STAMP(i) do { } while (0)
#endif

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

// ---- compile-time block sizes NX, NU <= 15: FOUR knots per wavefront, registers instead of LDS ----
// The kernel above pays two LDS reads per fma and a pair of wave syncs per pivot step of a tableau it walks with runtime
// indices: 1.44 ms for the 131072 rows of the BASELINE batch (1024 x 128, nx 14, nu 7), 0.5 TB/s.  Here a wavefront WALKS along
// a run of consecutive knots of one problem, four at a time, one knot per 16-lane quarter:
//   * lane l < NX of a quarter owns column l of Q_j (then of Q_j^-1, W_j = A_j Q_j^-1, T_j = W_j A_j' + V_j B_j'), lane l < NU
//     column l of R_j (R_j^-1, V_j = B_j R_j^-1) in registers with static indices; the spare lane NX carries q_j as one more column
//     through the elimination and the product (-> Q_j^-1 q_j, A_j Q_j^-1 q_j), lane NU carries r_j: the vectors cost nothing;
//   * the elimination is the in-place Gauss-Jordan of pinv_diag_quad_kernel (pinv.hip), Q and R steps interleaved so that both
//     share one LDS round trip per step (the pivot column is broadcast through a 16-element LDS line per quarter);
//   * operands a whole quarter needs (columns of A_j, B_j, W_j, V_j) are broadcast reads of LDS;
//   * every Q_j is inverted ONCE: T_j, W_j and the vector A e_j + B f_j of knot j stay in LDS slots for the quarter (or, across a
//     step, the carry slot) that builds row j+1 from them.  A run that does not start at knot 0 first runs one step on the four
//     knots before it with the stores switched off.  With one run per problem (batch >= 1024) nothing is computed twice;
//   * the inputs of the NEXT step (4 x 574 elements: G, C, g, c of four knots are contiguous) arrive by dword LDS-DMA while this
//     step computes -- no registers, no instructions beyond the 38 requests -- and all outputs leave as dense 256-byte stores
//     gathered from LDS (S rows [L_j | D_j | R_j] with L_j = -W_{j-1}, R_j = -W_j' read transposed; G^-1 in place of G).
// R_j and L_{j+1} are copies of the same registers, so S is exactly symmetric in storage as above.  Not bit-identical with
// the general kernel (no mirroring of the inverses, other summation order): both are held to the fp64 formulas by the tests.
namespace {

__device__ __forceinline__ void dma_dword(const void *base, uint32_t byte_off, uint32_t lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(byte_off), "s"(lds_addr) : "memory");
}
// Sixteen of them for 4 KB that are contiguous in memory and in LDS: the instruction offset moves both addresses, so one M0
// setting serves all (lane_off = 4 * lane).
#define GBDPCG_DMA_AT(o) "global_load_lds_dword %2, %1 offset:" #o "\n\t"
__device__ __forceinline__ void dma_4k(const void *base, uint32_t lane_off, uint32_t lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 GBDPCG_DMA_AT(0) GBDPCG_DMA_AT(256) GBDPCG_DMA_AT(512) GBDPCG_DMA_AT(768)
                 GBDPCG_DMA_AT(1024) GBDPCG_DMA_AT(1280) GBDPCG_DMA_AT(1536) GBDPCG_DMA_AT(1792)
                 GBDPCG_DMA_AT(2048) GBDPCG_DMA_AT(2304) GBDPCG_DMA_AT(2560) GBDPCG_DMA_AT(2816)
                 GBDPCG_DMA_AT(3072) GBDPCG_DMA_AT(3328) GBDPCG_DMA_AT(3584) GBDPCG_DMA_AT(3840)
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "v"(lane_off), "s"(lds_addr) : "memory");
}
#undef GBDPCG_DMA_AT
// One kilobyte per instruction (16 bytes per lane, lane16 = 16 * lane): global addresses need only be 4-byte aligned, like any
// dwordx4 load; the LDS side (M0 + 16 * lane) is 16-byte aligned by construction of the buffers.
#ifndef GBDPCG_SCHUR_PACKED
#define GBDPCG_SCHUR_PACKED 1   // 0: one row per instruction in the elimination (A/B runs)
#endif
#ifndef GBDPCG_SCHUR_SINGLE
#define GBDPCG_SCHUR_SINGLE 1   // 1: ONE input buffer, requested at the end of a step, D staged in the consumed A / B region: under 20 KB of LDS
#endif                          //    per wave in fp32 -> two waves per SIMD cover each other's waits; 0: two buffers, the next step's inputs
                                //    requested at the top of this one (one wave per SIMD)
#ifndef GBDPCG_SCHUR_NT
#define GBDPCG_SCHUR_NT 1       // 0: default cache policy for the 16-byte stores of S and G^-1 (A/B runs)
#endif
#ifndef GBDPCG_SCHUR_ST4
#define GBDPCG_SCHUR_ST4 1      // 0: one element per lane and store in the write-outs (A/B runs)
#endif
#ifndef GBDPCG_SCHUR_DMA_X4
#define GBDPCG_SCHUR_DMA_X4 1   // 0: dword requests only (A/B runs)
#endif
template <int COUNT> __device__ __forceinline__ void dma_x4(const void *base, uint32_t lane16, uint32_t lds_addr)
{
    static_assert(COUNT >= 1 && COUNT <= 4, "instruction offsets are 13 bits, signed");
    unsigned keep;
    if constexpr (COUNT == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "v"(lane16), "s"(lds_addr) : "memory");
    else if constexpr (COUNT == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "v"(lane16), "s"(lds_addr) : "memory");
    else if constexpr (COUNT == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:1024\n\tglobal_load_lds_dwordx4 %2, %1 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "v"(lane16), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:1024\n\tglobal_load_lds_dwordx4 %2, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "v"(lane16), "s"(lds_addr) : "memory");
}
// A region of DWORDS dwords (LDS room: the next multiple of 64), every lane of every instruction active: the caller has checked
// that the bytes up to the padded end exist in memory.  Whole kilobytes go as 16-byte requests, the rest as dwords.
template <uint32_t DWORDS> __device__ __forceinline__ void dma_region(const void *base, uint32_t lane_off, uint32_t lds_addr)
{
    constexpr uint32_t X4 = GBDPCG_SCHUR_DMA_X4 ? DWORDS / 256 : 0;   // instructions of 1 KB
    constexpr uint32_t INSTR = (DWORDS - X4 * 256 + 63) / 64;        // dword instructions behind them
#pragma unroll
    for (uint32_t i = 0; i + 4 <= X4; i += 4) dma_x4<4>(static_cast<const char *>(base) + i * 1024, lane_off * 4, lds_addr + i * 1024);
    if constexpr (X4 % 4 != 0) dma_x4<X4 % 4>(static_cast<const char *>(base) + (X4 / 4 * 4) * 1024, lane_off * 4, lds_addr + (X4 / 4 * 4) * 1024);
    const char *rest = static_cast<const char *>(base) + X4 * 1024;
    const uint32_t lrest = lds_addr + X4 * 1024;
#pragma unroll
    for (uint32_t i = 0; i + 16 <= INSTR; i += 16) dma_4k(rest + i * 256, lane_off, lrest + i * 256);
#pragma unroll
    for (uint32_t i = INSTR / 16 * 16; i < INSTR; ++i) dma_dword(rest + i * 256, lane_off, lrest + i * 256);
}

// 1 / x: the hardware reciprocal and one Newton step in fp32 (the division sequence is a dozen dependent instructions on the
// critical path of every pivot step), the division in fp64.
__device__ __forceinline__ float quad_rcp(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fma_t(fma_t(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double quad_rcp(double x) { return 1.0 / x; }

// The value lane J of every 16-lane row holds, in all lanes of that row: the DPP control row_newbcast (gfx90a and later) --
// a VALU move, no trip through the LDS crossbar (ds_swizzle in bit-mask mode does the same at an LDS instruction's cost:
// 294 of them per step and wave, on a pipe the four waves of a compute unit share).
template <int J> __device__ __forceinline__ float row_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, true));
}
template <int J> __device__ __forceinline__ double row_bcast(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + J, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + J, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// acc += coef[q] * (v of lane q of the row), q = Q0 .. QN-1 in ascending order: a dot product whose vector sits one entry per lane.
template <int Q0, int QN, int M, typename T> __device__ __forceinline__ void recover_dot(T &acc, const T (&coef)[M], T v)
{
    if constexpr (Q0 < QN) {
        acc = fma_t(coef[Q0], row_bcast<Q0>(v), acc);
        recover_dot<Q0 + 1, QN, M>(acc, coef, v);
    }
}
// acc[r] += (src[r] of lane q of the row) * coef[q] for q in [Q0, QN): a block product whose left factor lives one column per
// lane and whose right factor's column this lane holds in coef -- the broadcast rides on the fma as a DPP operand.
// (v_fmac_*_dpp written out: hipcc pairs the fmas into v_pk_fma_f32, which takes no DPP operand, and keeps a v_mov_b32_dpp per
// element next to them.  A VGPR a DPP operand reads must not have been written by the two preceding VALU instructions;
// the compiler does not look into asm for that, so every chain starts behind an s_nop and reads registers no instruction of
// the chain writes.)
template <int J> __device__ __forceinline__ void fmac_bcast(float &acc, float src, float coef)
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef), "n"(J));
}
template <int J> __device__ __forceinline__ void fmac_bcast(double &acc, double src, double coef)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef), "n"(J));
}
template <int Q0, int QN, int M, int K, typename T>
__device__ __forceinline__ void bcast_mac_chain(const T (&src)[M], const T (&coef)[K], T (&acc)[M])
{
    if constexpr (Q0 < QN) {
#pragma unroll
        for (int r = 0; r < M; ++r) fmac_bcast<Q0>(acc[r], src[r], coef[Q0]);
        bcast_mac_chain<Q0 + 1, QN, M, K>(src, coef, acc);
    }
}
template <int Q0, int QN, int M, int K, typename T>
__device__ __forceinline__ void bcast_mac(const T (&src)[M], const T (&coef)[K], T (&acc)[M])
{
    asm volatile("s_nop 1");
    bcast_mac_chain<Q0, QN, M, K>(src, coef, acc);
}

// One pivot step of the in-place Gauss-Jordan elimination on an M x M block held one column per lane (pinv_diag_quad_kernel's
// arithmetic): J is the pivot.
template <int J, int M, typename T> __device__ __forceinline__ void quad_pivot(T (&col)[M], uint32_t l)
{
    T cj[M];
#pragma unroll
    for (int r = 0; r < M; ++r) cj[r] = row_bcast<J>(col[r]);
    const bool is_j = l == (uint32_t)J;
    const T piv = quad_rcp(cj[J]);
    const T pr = is_j ? piv : col[J] * piv;
#if GBDPCG_SCHUR_PACKED
    if constexpr (sizeof(T) == 4) {
        // two rows per instruction: the pivot lane's "start from zero" is a packed multiply by 0 or 1 (exact) instead of a select
        // per row, the update a packed fma -- the same fma as below, element for element
        typedef float f2 __attribute__((ext_vector_type(2)));
        const float keep = is_j ? 0.0f : 1.0f;
        const f2 kk = {keep, keep}, npr = {-pr, -pr};
#pragma unroll
        for (int r = 0; r + 1 < M; r += 2) {
            const f2 c = {col[r], col[r + 1]}, b = {cj[r], cj[r + 1]};
            const f2 v = __builtin_elementwise_fma(b, npr, c * kk);
            col[r] = v.x;
            col[r + 1] = v.y;
        }
        if constexpr (M & 1) col[M - 1] = fma_t(-cj[M - 1], pr, is_j ? T(0) : col[M - 1]);
        col[J] = pr;
        return;
    }
#endif
#pragma unroll
    for (int r = 0; r < M; ++r) col[r] = (r == J) ? pr : fma_t(-cj[r], pr, is_j ? T(0) : col[r]);
}
// between(J) runs after pivot J: the caller spreads the previous step's stores over the elimination with it.
template <int J, int NX, int NU, typename T, typename F>
__device__ __forceinline__ void quad_eliminate(T (&Qc)[NX], T (&Rc)[NU], uint32_t l, F &&between)
{
    if constexpr (J < NX) {
        quad_pivot<J, NX>(Qc, l);
        if constexpr (J < NU) quad_pivot<J, NU>(Rc, l);
        between(J);
        quad_eliminate<J + 1, NX, NU>(Qc, Rc, l, between);
    }
}

template <typename T, int NX, int NU> struct QuadGeom {
    static constexpr uint32_t CP = 16;  // padded column
    static constexpr uint32_t SG = NX * NX + NU * NU, SC = NX * NX + NX * NU, SV = NX + NU;
    // raw inputs of one step (four knots), as they lie in memory
    // (each region padded to whole 256-byte DMA instructions)
    static constexpr uint32_t DW = sizeof(T) / 4;
    static constexpr uint32_t pad(uint32_t elems) { return (elems * DW + 63) / 64 * 64 / DW; }
    static constexpr uint32_t RG = 0, RC = RG + pad(4 * SG), Rg = RC + pad(4 * SC), Rc = Rg + pad(4 * SV), RAW_P = Rc + pad(4 * NX);
    static constexpr bool SINGLE = GBDPCG_SCHUR_SINGLE != 0;   // (fp64: 38 KB even so -- one wave per SIMD, but on every SIMD: with two buffers, 65 KB, half of them idle)
    static constexpr uint32_t WSL = (SINGLE ? 1 : 2) * RAW_P;     // 5 slots of (NX+1) padded columns: -[W_j | A e_j]
    static constexpr uint32_t TSL = WSL + 5 * (NX + 1) * CP;      // 5 slots of NX padded columns: T_j
    static constexpr uint32_t VSL = TSL + 5 * NX * CP;            // 5 slots of one padded column: B f_j
    // 4 x NX*NX: D_j, unpadded column-major -- with one input buffer in the A / B region of the inputs, which the products have
    // consumed by the time D is formed (pad(4 SC) >= 4 NX^2) and which the next request overwrites only after the write-out has
    // read it
    static constexpr uint32_t DSL = SINGLE ? RC : VSL + 5 * CP;
    static constexpr uint32_t GAM = SINGLE ? VSL + 5 * CP : DSL + 4 * NX * NX;
    static constexpr uint32_t ZER = (GAM + 4 * NX + 3) & ~3u;      // CP zeros: the "columns" of the lanes that own none
    static constexpr uint32_t TOTAL = ZER + CP;
    static constexpr uint32_t SROW = 3 * NX * NX;
};

}  // namespace

template <typename T, int NX, int NU>
__device__ __forceinline__ void schur_form_quad_body(uint32_t N, uint32_t run, uint32_t waves, const T *__restrict__ G,
                                                     const T *__restrict__ C, const T *__restrict__ g, const T *__restrict__ c,
                                                     T *__restrict__ S, T *__restrict__ gamma, T *__restrict__ Ginv)
{
    using Q = QuadGeom<T, NX, NU>;
    static_assert(NX <= 15 && NU <= 15 && NU <= NX, "one knot per 16-lane quarter, one spare lane for the vector");
    constexpr uint32_t CP = Q::CP, DW = sizeof(T) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw);
    uint32_t lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    __builtin_assume(lane < 64);   // or every "trip * 64 + lane < size" below is a compare, an exec mask and a branch
    const uint32_t l = lane & 15u, qd = lane >> 4;
    const uint32_t w = blockIdx.x;
    if (w >= waves) return;
    const uint32_t rpp = N / run;                 // runs per problem
    const uint32_t prob = w / rpp, j_start = (w - prob * rpp) * run;
    const KktDims d(NX, NU, N);
    const T *Gp = G + (size_t)prob * d.szG, *Cp = C + (size_t)prob * d.szC, *gp = g + (size_t)prob * d.szg, *cp = c + (size_t)prob * d.szc;

    // where the elements this lane stores in the S write-out sit in LDS (the same every step)
    // The write-outs move SW consecutive elements per lane and store (one 16-byte store in fp32: a store instruction costs this
    // kernel ~70 cycles of issue whatever its width -- 54 dword stores per step were a quarter of its time): OUT_T trips for the
    // four S rows of a step, GI_T for the G^-1 of its four knots; ST_I store instructions per trip.
    constexpr uint32_t SW = GBDPCG_SCHUR_ST4 && Q::SROW % 4 == 0 ? 4 : 1;   // (an odd block size: rows of 3 n^2 elements do not split into fours)
    constexpr uint32_t ST_I = SW * sizeof(T) > 16 ? SW * sizeof(T) / 16 : 1;
    constexpr uint32_t OUT_T = (4 * Q::SROW / SW + 63) / 64, GI_T = (4 * Q::SG / SW + 63) / 64;
    constexpr uint32_t S_STORES = OUT_T * ST_I + 1, GI_STORES = GI_T * ST_I;   // + 1: gamma
    static_assert((4 * Q::SROW) % SW == 0 && Q::SROW % SW == 0 && (4 * Q::SG) % SW == 0, "whole groups");
    typedef T OutV __attribute__((ext_vector_type(SW == 4 ? 4 : 2), aligned(sizeof(T))));   // (SW == 1 does not use it)
    uint32_t src[OUT_T][SW];
#pragma unroll
    for (uint32_t t = 0; t < OUT_T; ++t)
#pragma unroll
        for (uint32_t u = 0; u < SW; ++u) {
            const uint32_t e0 = (t * 64 + lane) * SW + u, e = e0 < 4 * Q::SROW ? e0 : 0u;
            const uint32_t q = e / Q::SROW, i = e - q * Q::SROW, slot = i / (NX * NX), ii = i - slot * (NX * NX), cc = ii / NX, r = ii - cc * NX;
            src[t][u] = slot == 0 ? Q::WSL + q * (NX + 1) * CP + cc * CP + r
                        : slot == 1 ? Q::DSL + q * NX * NX + ii
                                    : Q::WSL + (q + 1) * (NX + 1) * CP + r * CP + cc;
        }
    // SW elements from registers to memory (4-byte aligned addresses are fine for a 16-byte store, as for the loads)
    auto put = [&](T *dst, const T (&v)[SW]) {
        if constexpr (SW == 1) {
            dst[0] = v[0];
        } else {
            OutV o;
#pragma unroll
            for (uint32_t u = 0; u < SW; ++u) o[u] = v[u];
#if GBDPCG_SCHUR_NT
            __builtin_nontemporal_store(o, reinterpret_cast<OutV *>(dst));
#else
            *reinterpret_cast<OutV *>(dst) = o;
#endif
        }
    };
    // carry slots of a run that starts a problem: L_0 = 0, D_0 = Q_0^-1, gamma_0 = -(c_0 + Q_0^-1 q_0)
    // (written into the last quarter's slots: every step begins by moving those into slot 0)
    for (uint32_t i = lane; i < (NX + 1) * CP; i += 64) lds[Q::WSL + 4 * (NX + 1) * CP + i] = T(0);
    for (uint32_t i = lane; i < NX * CP; i += 64) lds[Q::TSL + 4 * NX * CP + i] = T(0);
    if (lane < CP) lds[Q::VSL + 4 * CP + lane] = T(0);
    if (lane < CP) lds[Q::ZER + lane] = T(0);

    // requests for the four knots from jb on into raw buffer b (elements past the end of the problem's arrays are not requested)
    const uint32_t batch = waves / rpp;
    auto request = [&](uint32_t jb, uint32_t b) {
        const uint32_t base = (uint32_t)(uintptr_t)(lds + b * Q::RAW_P);
        {
            // whole instructions while the bytes up to each padded end lie inside the arrays (what is read past a problem's own
            // end -- the R, A, B, r its last knot does not have -- is the next problem's data and is overwritten below)
            const size_t eG = (size_t)prob * d.szG + (size_t)jb * Q::SG + Q::pad(4 * Q::SG), eC = (size_t)prob * d.szC + (size_t)jb * Q::SC + Q::pad(4 * Q::SC);
            const size_t eg = (size_t)prob * d.szg + (size_t)jb * Q::SV + Q::pad(4 * Q::SV), ec = (size_t)prob * d.szc + (size_t)jb * NX + Q::pad(4 * NX);
            if (eG <= (size_t)batch * d.szG && eC <= (size_t)batch * d.szC && eg <= (size_t)batch * d.szg && ec <= (size_t)batch * d.szc) {
                const uint32_t lo = lane * 4;
                dma_region<4 * Q::SG * DW>(Gp + (size_t)jb * Q::SG, lo, base + Q::RG * DW * 4);
                dma_region<4 * Q::SC * DW>(Cp + (size_t)jb * Q::SC, lo, base + Q::RC * DW * 4);
                dma_region<4 * Q::SV * DW>(gp + (size_t)jb * Q::SV, lo, base + Q::Rg * DW * 4);
                dma_region<4 * NX * DW>(cp + (size_t)jb * NX, lo, base + Q::Rc * DW * 4);
                return;
            }
        }
        const uint32_t lim_G = (uint32_t)(d.szG - (size_t)jb * Q::SG) * DW, lim_C = jb < N - 1 ? (uint32_t)(d.szC - (size_t)jb * Q::SC) * DW : 0u;
        const uint32_t lim_g = (uint32_t)(d.szg - (size_t)jb * Q::SV) * DW;
        const T *sG = Gp + (size_t)jb * Q::SG, *sC = Cp + (size_t)jb * Q::SC, *sg = gp + (size_t)jb * Q::SV, *sc = cp + (size_t)jb * NX;
#pragma unroll
        for (uint32_t o = 0; o < 4 * Q::SG * DW; o += 64)
            if (o + lane < 4 * Q::SG * DW && o + lane < lim_G) dma_dword(sG, (o + lane) * 4, base + (Q::RG * DW + o) * 4);
#pragma unroll
        for (uint32_t o = 0; o < 4 * Q::SC * DW; o += 64)
            if (o + lane < 4 * Q::SC * DW && o + lane < lim_C) dma_dword(sC, (o + lane) * 4, base + (Q::RC * DW + o) * 4);
#pragma unroll
        for (uint32_t o = 0; o < 4 * Q::SV * DW; o += 64)
            if (o + lane < 4 * Q::SV * DW && o + lane < lim_g) dma_dword(sg, (o + lane) * 4, base + (Q::Rg * DW + o) * 4);
        const uint32_t lim_c = (uint32_t)(d.szc - (size_t)jb * NX) * DW;
#pragma unroll
        for (uint32_t o = 0; o < 4 * NX * DW; o += 64)
            if (o + lane < 4 * NX * DW && o + lane < lim_c) dma_dword(sc, (o + lane) * 4, base + (Q::Rc * DW + o) * 4);
    };

    const bool pre = j_start != 0;                     // one silent step on the four knots before the run
    const uint32_t j_first = pre ? j_start - 4 : j_start, j_end = j_start + run;
    request(j_first, 0);
    uint32_t b = 0, stores_since_request = 0;
    bool pending = false;   // S / gamma of the previous step still sit in LDS
    T *So_prev = S, *gam_prev = gamma;
#ifdef GBDPCG_SCHUR_STAMPS
    unsigned long long st[12] = {};
#endif
    constexpr bool SINGLE = Q::SINGLE;
    T outv[OUT_T][SW], outg = T(0);   // S rows and gamma of the previous step: read at its end, stored during this step's elimination
#pragma unroll
    for (uint32_t t = 0; t < OUT_T; ++t)
#pragma unroll
        for (uint32_t u = 0; u < SW; ++u) outv[t][u] = T(0);
    for (uint32_t jb = j_first; jb < j_end; jb += 4, b ^= (SINGLE ? 0u : 1u)) {
        const bool emit = jb >= j_start;
#ifdef GBDPCG_SCHUR_STAMPS
        const bool stamp_now = blockIdx.x == 0 && jb == 20;
#endif
        SCHUR_STAMP(0);
        // The requests for this step were issued at the top of the previous one, before every store that step issued (the
        // deferred S / gamma stores of the step before it: S_STORES instructions, and its own G^-1 stores: GI_STORES -- a known number per
        // instruction per trip, which is why the write-out loops are written trip by trip), and the memory operations of a wave
        // retire in order: waiting until exactly that many are left is waiting for the requests and for nothing else.
        // (One input buffer: the requests were the last thing the previous step issued, so everything is waited for.)
        static_assert(S_STORES + GI_STORES <= 63, "vmcnt is a 6-bit counter");
        switch ((GBDPCG_SCHUR_SKIP & 4) || SINGLE ? 0u : stores_since_request) {
        case S_STORES + GI_STORES: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(S_STORES + GI_STORES) : "memory"); break;
        case S_STORES: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(S_STORES) : "memory"); break;
        case GI_STORES: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GI_STORES) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
        stores_since_request = 0;
        SCHUR_STAMP(1);
        if (!SINGLE && jb + 4 < j_end && !(GBDPCG_SCHUR_SKIP & 8)) request(jb + 4, b ^ 1u);
        SCHUR_STAMP(2);
        T *raw = lds + b * Q::RAW_P;
        T *rQ = raw + Q::RG + qd * Q::SG, *rR = rQ + NX * NX, *rA = raw + Q::RC + qd * Q::SC, *rB = rA + NX * NX;
        if (jb + 4 >= N) {  // the last knot has no R, A, B, r in memory: R = I, the rest 0 (W = V = 0: R_{N-1} = 0)
            const uint32_t ql = N - 1 - jb;   // its quarter; knotPoints % 4 != 0 leaves quarters behind it that own no knot
            T *lR = raw + Q::RG + ql * Q::SG + NX * NX, *lA = raw + Q::RC + ql * Q::SC, *lr = raw + Q::Rg + ql * Q::SV + NX;
            for (uint32_t i = lane; i < NU * NU; i += 64) lR[i] = (i / NU == i % NU) ? T(1) : T(0);
            for (uint32_t i = lane; i < Q::SC; i += 64) lA[i] = T(0);
            if (lane < NU) lr[lane] = T(0);
            // quarters without a knot eliminate the identity (what their buffers hold is stale or the next problem's; nothing
            // they compute is stored or read by a quarter that owns a knot)
            for (uint32_t q = ql + 1; q < 4; ++q) {
                T *dQ = raw + Q::RG + q * Q::SG, *dR = dQ + NX * NX, *dA = raw + Q::RC + q * Q::SC;
                for (uint32_t i = lane; i < NX * NX; i += 64) dQ[i] = (i / NX == i % NX) ? T(1) : T(0);
                for (uint32_t i = lane; i < NU * NU; i += 64) dR[i] = (i / NU == i % NU) ? T(1) : T(0);
                for (uint32_t i = lane; i < Q::SC; i += 64) dA[i] = T(0);
            }
        }
        wave_sync();
        // this lane's columns: Q_j (lane NX: q_j), R_j (lane NU: r_j)
        T Qc[NX], Rc[NU];
        {
            const T *zero = lds + Q::ZER;
            const T *pq = l < NX ? rQ + l * NX : (l == NX ? raw + Q::Rg + qd * Q::SV : zero);
            const T *pr = l < NU ? rR + l * NU : (l == NU ? raw + Q::Rg + qd * Q::SV + NX : zero);
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) Qc[r] = pq[r];
#pragma unroll
            for (uint32_t r = 0; r < NU; ++r) Rc[r] = pr[r];
        }
        SCHUR_STAMP(3);
        // S and gamma of the PREVIOUS step leave now, a few stores after every pivot: all waves of the device walk in step, and
        // stores issued in one piece at the end of a step reach the memory system as one burst (14 MB) that the next requests
        // then queue behind
        constexpr uint32_t PER = (OUT_T + NX - 2) / (NX - 1);
        static_assert(PER * (NX - 1) >= OUT_T, "the last pivot's slot is gamma's");
        if (pending) stores_since_request += S_STORES;
        auto drain = [&](uint32_t J) {
#pragma unroll
            for (uint32_t t = J * PER; t < (J + 1) * PER && t < OUT_T; ++t)
                if (J + 1 < NX && (t * 64 + lane) * SW < 4 * Q::SROW) put(So_prev + (t * 64 + lane) * SW, outv[t]);
            if (J + 1 == NX && lane < 4 * NX) gam_prev[lane] = outg;
        };
        if (!(GBDPCG_SCHUR_SKIP & 1)) {
            if (pending && !SINGLE) quad_eliminate<0, NX, NU>(Qc, Rc, l, drain);
            else quad_eliminate<0, NX, NU>(Qc, Rc, l, [](uint32_t) {});
        } else if (pending && !SINGLE) {
#pragma unroll
            for (uint32_t J = 0; J < NX; ++J) drain(J);
        }
        SCHUR_STAMP(4);
        // carry: the last quarter's slots of the previous step become slot 0 of this one (reads first, then writes: the
        // compiler must assume that one LDS write changes what the next LDS read sees and will not batch them itself)
        {
            constexpr uint32_t WT = ((NX + 1) * CP + 63) / 64, TT = (NX * CP + 63) / 64;
            T cw[WT], ct[TT];
#pragma unroll
            for (uint32_t t = 0; t < WT; ++t) cw[t] = lds[Q::WSL + 4 * (NX + 1) * CP + (t * 64 + lane < (NX + 1) * CP ? t * 64 + lane : 0u)];
#pragma unroll
            for (uint32_t t = 0; t < TT; ++t) ct[t] = lds[Q::TSL + 4 * NX * CP + (t * 64 + lane < NX * CP ? t * 64 + lane : 0u)];
            const T cv = lds[Q::VSL + 4 * CP + (lane & (CP - 1))];
#pragma unroll
            for (uint32_t t = 0; t < WT; ++t)
                if (t * 64 + lane < (NX + 1) * CP) lds[Q::WSL + t * 64 + lane] = cw[t];
#pragma unroll
            for (uint32_t t = 0; t < TT; ++t)
                if (t * 64 + lane < NX * CP) lds[Q::TSL + t * 64 + lane] = ct[t];
            if (lane < CP) lds[Q::VSL + lane] = cv;
        }
        wave_sync();
        SCHUR_STAMP(5);
        // G^-1 in place of G (the write-out copies the region)
        if (Ginv) {
            if (l < NX) {
#pragma unroll
                for (uint32_t r = 0; r < NX; ++r) rQ[l * NX + r] = Qc[r];
            }
            if (l < NU) {
#pragma unroll
                for (uint32_t r = 0; r < NU; ++r) rR[l * NU + r] = Rc[r];
            }
        }
        // [W | A e] = A [Q^-1 | e],  [V | B f] = B [R^-1 | f],  T = W A' + V B': the left factors one column per lane in
        // registers (A, B picked up from LDS once; W, V where they were computed), broadcast lane by lane inside the fma
        T Ac[NX], Bc[NX], ar[NX], brow[NU];
        {
            const uint32_t lc = l < NX ? l : 0u, lb = l < NU ? l : 0u;
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) {
                Ac[r] = rA[lc * NX + r];
                Bc[r] = rB[lb * NX + r];
                ar[r] = rA[r * NX + lc];            // row l of A
            }
#pragma unroll
            for (uint32_t q = 0; q < NU; ++q) brow[q] = rB[q * NX + lc];  // row l of B
        }
        T Xc[NX], Yc[NX], Tc[NX];
#pragma unroll
        for (uint32_t r = 0; r < NX; ++r) Xc[r] = Yc[r] = Tc[r] = T(0);
        constexpr int PX = (GBDPCG_SCHUR_SKIP & 2) ? 1 : NX, PU = (GBDPCG_SCHUR_SKIP & 2) ? 1 : NU;
        bcast_mac<0, PX, NX, NX>(Ac, Qc, Xc);
        bcast_mac<0, PU, NX, NU>(Bc, Rc, Yc);
        T *wq = lds + Q::WSL + (qd + 1) * (NX + 1) * CP, *vq = lds + Q::VSL + (qd + 1) * CP, *tq = lds + Q::TSL + (qd + 1) * NX * CP;
        if (l <= NX) {
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) wq[l * CP + r] = T(0) - Xc[r];
        }
        if (l == NU) {
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) vq[r] = Yc[r];
        }
        SCHUR_STAMP(6);
        bcast_mac<0, PX, NX, NX>(Xc, ar, Tc);
        bcast_mac<0, PU, NX, NU>(Yc, brow, Tc);
        if (l < NX) {
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) tq[l * CP + r] = Tc[r];
        }
        wave_sync();
        SCHUR_STAMP(7);
        // D_j = T_{j-1} + Q_j^-1;  gamma_j = -(c_j + e_j - (A e + B f)_{j-1})
        const T *tp = tq - NX * CP, *wp = wq - (NX + 1) * CP, *vp = vq - CP;
        {   // (reads first, then writes, as above)
            T tv[NX], cv[NX], wv[NX], vv[NX];
            const uint32_t lc = l < NX ? l : 0u;
#pragma unroll
            for (uint32_t r = 0; r < NX; ++r) {
                tv[r] = tp[lc * CP + r];
                cv[r] = raw[Q::Rc + qd * NX + r];
                wv[r] = wp[NX * CP + r];
                vv[r] = vp[r];
            }
            if (l < NX) {
#pragma unroll
                for (uint32_t r = 0; r < NX; ++r) lds[Q::DSL + qd * NX * NX + l * NX + r] = tv[r] + Qc[r];
            }
            if (l == NX) {
#pragma unroll
                for (uint32_t r = 0; r < NX; ++r) lds[Q::GAM + qd * NX + r] = -(cv[r] + Qc[r] - (vv[r] - wv[r]));
            }
        }
        wave_sync();
        SCHUR_STAMP(8);
        pending = emit && !((GBDPCG_SCHUR_SKIP & 4) && jb != 0);
        if (pending) {
            So_prev = S + ((size_t)prob * N + jb) * Q::SROW;
            gam_prev = gamma + ((size_t)prob * N + jb) * NX;
            if (Ginv) {
                T *Go = Ginv + (size_t)prob * d.szG + (size_t)jb * Q::SG;
                const uint32_t lim = (uint32_t)(d.szG - (size_t)jb * Q::SG);
                if (lim >= 4 * Q::SG) {   // every step but a problem's last: whole trips, the reads in one batch
                    T gv[GI_T][SW];
#pragma unroll
                    for (uint32_t t = 0; t < GI_T; ++t)
#pragma unroll
                        for (uint32_t u = 0; u < SW; ++u) gv[t][u] = raw[Q::RG + (t * 64 + lane) * SW + u];
#pragma unroll
                    for (uint32_t t = 0; t < GI_T; ++t)
                        if ((t * 64 + lane) * SW < 4 * Q::SG) put(Go + (t * 64 + lane) * SW, gv[t]);
                } else {   // (no request follows a problem's last step: the count below is not looked at again)
                    for (uint32_t i = lane; i < lim; i += 64) Go[i] = raw[Q::RG + i];
                }
                stores_since_request += GI_STORES;
            }
            // the S rows and gamma of this step, in one round trip, before their slots (and, with one input buffer, the inputs) are
            // touched again
#pragma unroll
            for (uint32_t t = 0; t < OUT_T; ++t)
#pragma unroll
                for (uint32_t u = 0; u < SW; ++u) outv[t][u] = lds[src[t][u]];
            outg = lds[Q::GAM + (lane < 4 * NX ? lane : 0u)];
            if constexpr (SINGLE) {
                // two waves per SIMD are out of step with each other: the rows leave at once (no 40 registers held through the
                // next step's elimination, which is what lets two waves fit)
                const uint32_t live = jb + 4 > N ? N - jb : 4u;
#pragma unroll
                for (uint32_t t = 0; t < OUT_T; ++t)
                    if ((t * 64 + lane) * SW < live * Q::SROW) put(So_prev + (t * 64 + lane) * SW, outv[t]);
                if (lane < live * NX) gam_prev[lane] = outg;
                pending = false;
            }
        }
        if (SINGLE && jb + 4 < j_end && !(GBDPCG_SCHUR_SKIP & 8)) {
            // nothing orders an LDS read behind an LDS-DMA write: every read of the buffer has returned before the requests go out
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            request(jb + 4, 0);
        }
#ifdef GBDPCG_SCHUR_STAMPS
        {
            const bool stamp_now = blockIdx.x == 0 && jb == 20;
            SCHUR_STAMP(9);
        }
#endif
    }
    // S and gamma of the last step (the last step of a problem whose knotPoints are not a multiple of 4 holds fewer than 4 rows)
    if (pending) {
        const uint32_t live = j_end == N && (N & 3u) ? (N & 3u) : 4u;
#pragma unroll
        for (uint32_t t = 0; t < OUT_T; ++t)
            if ((t * 64 + lane) * SW < live * Q::SROW) put(So_prev + (t * 64 + lane) * SW, outv[t]);
        if (lane < live * NX) gam_prev[lane] = outg;
    }
#ifdef GBDPCG_SCHUR_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *out = reinterpret_cast<unsigned long long *>(gamma + (size_t)(waves / rpp) * N * NX);  // 96 bytes past the end: the tool allocates them
        for (int i = 0; i < 12; ++i) out[i] = st[i];
    }
#endif
}

template <typename T, int NX, int NU>
__global__ __launch_bounds__(64) void schur_form_quad_kernel(uint32_t N, uint32_t run, uint32_t waves, const T *__restrict__ G,
                                                            const T *__restrict__ C, const T *__restrict__ g,
                                                            const T *__restrict__ c, T *__restrict__ S, T *__restrict__ gamma,
                                                            T *__restrict__ Ginv)
{
    schur_form_quad_body<T, NX, NU>(N, run, waves, G, C, g, c, S, gamma, Ginv);
}
// The form with one input buffer (fp32): held to 256 registers so that two waves share a SIMD.
template <int NX, int NU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void schur_form_quad2_kernel(
    uint32_t N, uint32_t run, uint32_t waves, const float *__restrict__ G, const float *__restrict__ C, const float *__restrict__ g,
    const float *__restrict__ c, float *__restrict__ S, float *__restrict__ gamma, float *__restrict__ Ginv)
{
    schur_form_quad_body<float, NX, NU>(N, run, waves, G, C, g, c, S, gamma, Ginv);
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

// ---- compile-time block sizes NX, NU <= 16: FOUR knots per wavefront, no LDS at all ----
// The kernel above stages every block in LDS and walks it with runtime indices (two LDS reads per fma): 172 us for the 131072
// rows of the BASELINE batch, 1.8 TB/s, on a step that moves 2.4 KB per row and has 0.5 flop per byte.  Here a 16-lane quarter
// owns one row (problem, k) and every operand goes from memory straight into the registers of the lane that multiplies it:
//   * lane l holds COLUMN l of A_k and B_k (14 contiguous elements each: (A' lambda+)_l and (B' lambda+)_l are dot products along
//     a column) and ROW l of Q_k^-1 and R_k^-1 (element q of it comes with the quarter's q-th load: 14 lanes x 4 bytes, contiguous);
//   * the vector a product multiplies sits one entry per lane (lambda_{k+1}; then t_x, t_u where they were computed) and reaches
//     the fma as a DPP row broadcast -- no LDS, no shuffles through the crossbar;
//   * every load of a row is requested before the first fma (53 registers of operands per lane), five or six waves per SIMD keep
//     ~200 KB per compute unit in flight.
// Same sums in the same order as the kernel above (q ascending, one fma chain per output entry): bit-identical results.
template <typename T, int NX, int NU>
__global__ __launch_bounds__(256) void schur_recover_quad_kernel(uint32_t N, uint64_t rows, const T *__restrict__ Ginv,
                                                                const T *__restrict__ C, const T *__restrict__ g,
                                                                const T *__restrict__ lambda, T *__restrict__ z)
{
    static_assert(NX <= 16 && NU <= NX, "one row per 16-lane quarter");
    uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u;
    const uint64_t row = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < rows;
    const KktDims d(NX, NU, N);
    const uint64_t prob = live ? row / N : 0;
    const uint32_t k = live ? (uint32_t)(row - prob * N) : 0u;
    const bool has_next = live && k + 1 < N;
    const bool lx = live && l < NX, lu = has_next && l < NU;
    const uint32_t cx = l < NX ? l : 0u, cu = l < NU ? l : 0u;   // clamped: idle lanes read what a live lane reads
    const T *Gi = Ginv + prob * d.szG + (size_t)k * d.sg, *Ck = C + prob * d.szC + (size_t)k * d.sc;
    const T *gk = g + prob * d.szg + (size_t)k * d.sv;
    const T *lk = lambda + (prob * N + k) * NX;

    T a[NX], b[NX], qi[NX], ri[NU];
    T lam_n = T(0), tx = T(0), tu = T(0);
    if (live) {
        tx = gk[cx] + lk[cx];
#pragma unroll
        for (int q = 0; q < NX; ++q) qi[q] = Gi[q * NX + cx];
    } else {
#pragma unroll
        for (int q = 0; q < NX; ++q) qi[q] = T(0);
    }
    if (has_next) {
        lam_n = lk[NX + cx];
        tu = gk[NX + cu];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            a[q] = Ck[cx * NX + q];
            b[q] = Ck[NX * NX + cu * NX + q];
        }
#pragma unroll
        for (int q = 0; q < NU; ++q) ri[q] = Gi[NX * NX + q * NU + cu];
    } else {
#pragma unroll
        for (int q = 0; q < NX; ++q) a[q] = b[q] = T(0);
#pragma unroll
        for (int q = 0; q < NU; ++q) ri[q] = T(0);
    }
    // t_x = q_k + lambda_k - A_k' lambda_{k+1},  t_u = r_k - B_k' lambda_{k+1}   (the rows of the last knot have neither product)
    T sa = T(0), sb = T(0);
    recover_dot<0, NX>(sa, a, lam_n);
    recover_dot<0, NX>(sb, b, lam_n);
    if (has_next) {
        tx -= sa;
        tu -= sb;
    }
    // x_k = -Q_k^-1 t_x,  u_k = -R_k^-1 t_u
    T sx = T(0), su = T(0);
    recover_dot<0, NX>(sx, qi, tx);
    recover_dot<0, NU>(su, ri, tu);
    T *zk = z + prob * d.szg + (size_t)k * d.sv;
    if (lx) zk[l] = -sx;
    if (lu) zk[NX + l] = -su;
}

// Waves per workgroup for a per-wave LDS need; 0 = does not fit one CU.
static uint32_t waves_for(const DeviceInfo &dev, size_t wave_bytes)
{
    if (wave_bytes > dev.lds_per_wg_max) return 0;
    uint32_t w = 4;
    while (w > 1 && w * wave_bytes > 64 * 1024) --w;
    return w;
}

// The block sizes the four-knots-per-wave kernels are built for: stateSize = 2 x joints, controlSize = joints (a manipulator's
// positions and velocities against its torques; 14 / 7 is the BASELINE shape), and the pendulum (2 / 1), cart-pole (4 / 1) and
// quadrotor (12 / 4, 13 / 4 with a quaternion) shapes of the MPC literature.  Other sizes take the any-size LDS kernels.
#define GBDPCG_QUAD_SHAPES(X) X(2, 1) X(4, 1) X(4, 2) X(6, 3) X(8, 4) X(10, 5) X(12, 4) X(12, 6) X(13, 4) X(14, 7) \
    X(3, 1) X(5, 2) X(6, 1) X(6, 2) X(7, 3) X(8, 2) X(9, 3) X(10, 4) X(11, 4) X(12, 3)   /* round 3: under-actuated and odd shapes (9 / 3, 1024 x 128: formation 512 -> 95 us, recovery 95 -> 30 us) */

template <typename T, int NX, int NU>
hipError_t launch_form_quad(const DeviceInfo &dev, uint32_t N, uint32_t batch, const T *G, const T *C, const T *g, const T *c, T *S,
                            T *gamma, T *Ginv, hipStream_t s)
{
    using Q = QuadGeom<T, NX, NU>;
    // one run per problem when the batch alone fills the device, shorter runs (each pays one silent step) otherwise
    // (runs are multiples of 4 knots that divide knotPoints: other horizons are one run, the last step partly empty)
    uint32_t run = N;
    const size_t lds_wave = (size_t)Q::TOTAL * sizeof(T);
    const uint64_t fit = lds_wave ? (160u * 1024u) / lds_wave : 8u;   // waves of this kernel a compute unit's LDS holds
    const uint64_t want = (fit > 8 ? 8 : (fit < 1 ? 1 : fit)) * dev.num_cus;   // ... and the device, at two per SIMD at most
    while (run % 8 == 0 && (uint64_t)batch * (N / run) < want) run /= 2;
    const uint64_t nwaves = (uint64_t)batch * (N / run);
    if (nwaves > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)Q::TOTAL * sizeof(T);
    void (*kern)(uint32_t, uint32_t, uint32_t, const T *, const T *, const T *, const T *, T *, T *, T *) = schur_form_quad_kernel<T, NX, NU>;
    if constexpr (Q::SINGLE && sizeof(T) == 4) kern = schur_form_quad2_kernel<NX, NU>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)nwaves), dim3(64), lds, s, N, run, (uint32_t)nwaves, G, C, g, c, S, gamma, Ginv);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_form_schur(const DeviceInfo &dev, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *G, const T *C,
                             const T *g, const T *c, T *S, T *gamma, T *Ginv, hipStream_t s)
{
    // GBDPCG_SCHUR_GENERAL=1: the any-size kernel also where the four-knots-per-wave form exists (A/B runs, tests)
    const char *env = getenv("GBDPCG_SCHUR_GENERAL");
    if (!(env && env[0] == '1')) {
#define GBDPCG_X(NX, NU) \
    if (nx == NX && nu == NU) return launch_form_quad<T, NX, NU>(dev, N, batch, G, C, g, c, S, gamma, Ginv, s);
        GBDPCG_QUAD_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    }
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
    const uint64_t rows = (uint64_t)batch * N;
    // GBDPCG_SCHUR_GENERAL=1: the any-size kernel also where the four-rows-per-wave form exists (A/B runs, tests)
    const char *env = getenv("GBDPCG_SCHUR_GENERAL");
    if (!(env && env[0] == '1')) {
        const uint64_t grid = (rows + 15) / 16;   // 4 waves x 4 rows per workgroup
        if (grid > 0x7fffffffull) return hipErrorInvalidValue;
#define GBDPCG_X(NX, NU)                                                                                                              \
    if (nx == NX && nu == NU) {                                                                                                       \
        hipLaunchKernelGGL((schur_recover_quad_kernel<T, NX, NU>), dim3((uint32_t)grid), dim3(256), 0, s, N, rows, Ginv, C, g, lambda, z); \
        return hipGetLastError();                                                                                                     \
    }
        GBDPCG_QUAD_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    }
    const size_t wave_bytes = (size_t)recover_wave_elems(nx, nu) * sizeof(T);
    const uint32_t waves = waves_for(dev, wave_bytes);
    if (!waves) return hipErrorInvalidValue;
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
