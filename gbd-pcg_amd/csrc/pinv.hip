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
// One thread GROUP per (problem, knot): a whole 256-thread workgroup for large blocks, ONE WAVEFRONT
// (four knots per workgroup) when n <= 24 -- these blocks are so small (n = 14: 196 elements) that a
// knot is pure latency, and a wave needs no workgroup barrier: LDS operations of one wave execute in
// program order, so a compiler-level fence between "everyone has read the old tableau" and "write the
// new one" is all the synchronisation there is.  Pass 1 inverts D_k by Gauss-Jordan on an LDS-resident
// [D | I] tableau (no pivoting: D_k is a definite diagonal block of a Schur complement); pass 2
// (STAIR only) does the two triple products with all three operands in LDS.  O(n^3) per knot on n^2
// data, once per control step; not the bandwidth-bound hot loop, but on the MPC critical path.
#include "bt_device.hpp"
#include "internal.hpp"

#ifndef GBDPCG_PINV_SKIP
#define GBDPCG_PINV_SKIP 0     // timing builds of pinv_stair_mfma_kernel (WRONG results): 1 no elimination, 2 no products, 4 no stores
#endif
#ifndef GBDPCG_PINV_PACKED
#define GBDPCG_PINV_PACKED 1   // 0: one row per instruction in the DPP elimination (A/B runs)
#endif
#ifndef GBDPCG_PINV_DPP
#define GBDPCG_PINV_DPP 1   // 0: pivot columns of the one-launch stair kernel broadcast through LDS (A/B runs)
#endif

namespace gbdpcg {

constexpr int kPinvThreads = 256;

// Synchronise the GT threads that share one knot.
template <int GT> __device__ __forceinline__ void group_sync()
{
    if constexpr (GT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <typename T, int GT, int EPT_MAX>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_kernel(uint32_t n, uint32_t N, uint64_t knots,
                                                                const T *__restrict__ S, T *__restrict__ Pinv, int kind)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr uint32_t GROUPS = kPinvThreads / GT;
    const uint32_t grp = threadIdx.x / GT, tid = threadIdx.x % GT;
    const uint64_t knot = (uint64_t)blockIdx.x * GROUPS + grp;  // (problem, knot) flattened: same stride
    const uint32_t w = 2 * n;
    T *tab = reinterpret_cast<T *>(smem_raw) + (size_t)grp * align16<T>(2 * n * n);  // [n][2n] row-major tableau
    if (GT == 64 && knot >= knots) return;  // whole wave idle (only the wave-per-knot form has spare groups)
    const size_t blk = (size_t)knot * 3 * n * n;
    const T *D = S + blk + (size_t)n * n;
    T *out = Pinv + blk;

    for (uint32_t i = tid; i < n * n; i += GT) {
        const uint32_t c = i / n, r = i - c * n;  // column-major source
        tab[r * w + c] = (kind == 0) ? (r == c ? T(1) : T(0)) : D[i];
        tab[r * w + n + c] = (r == c) ? T(1) : T(0);
    }
    group_sync<GT>();
    if (kind != 0) {
        for (uint32_t j = 0; j < n; ++j) {
            // every thread first READS what it needs of the old tableau (pivot, its column-j entries, the
            // pivot row), then all WRITE: the two halves are separated by group_sync
            const T piv = T(1) / tab[j * w + j];
            T upd[EPT_MAX];
            uint32_t cnt = 0;
            for (uint32_t i = tid; i < n * w && cnt < EPT_MAX; i += GT, ++cnt) {
                const uint32_t r = i / w, c = i - r * w;
                const T pr = tab[j * w + c] * piv;  // scaled pivot-row entry
                upd[cnt] = (r == j) ? pr : fma_t(-tab[r * w + j], pr, tab[r * w + c]);
            }
            group_sync<GT>();
            cnt = 0;
            for (uint32_t i = tid; i < n * w && cnt < EPT_MAX; i += GT, ++cnt) tab[i] = upd[cnt];
            group_sync<GT>();
        }
    }
    // D_k^-1 of a symmetric D_k is symmetric in exact arithmetic but not bit for bit after the
    // elimination; the upper triangle is mirrored so that the stair blocks built from it come out
    // exactly symmetric (L_{k+1} == R_k^T) whenever S is -- what gbdpcg_set_symmetric relies on.
    for (uint32_t i = tid; i < n * n; i += GT) {
        const uint32_t c = i / n, r = i - c * n;
        out[(size_t)n * n + i] = r <= c ? tab[r * w + n + c] : tab[c * w + n + r];
        out[i] = T(0);
        out[(size_t)2 * n * n + i] = T(0);
    }
}

// Off-diagonal slots of the stair preconditioner; reads the D slots pass 1 wrote.
template <typename T, int GT>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_kernel(uint32_t n, uint32_t N, uint64_t knots,
                                                                 const T *__restrict__ S, T *Pinv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr uint32_t GROUPS = kPinvThreads / GT;
    const uint32_t grp = threadIdx.x / GT, tid = threadIdx.x % GT;
    const uint64_t knot = (uint64_t)blockIdx.x * GROUPS + grp;
    const uint32_t nn = n * n;
    T *A = reinterpret_cast<T *>(smem_raw) + (size_t)grp * align16<T>(4 * nn);  // first factor  (column-major)
    T *B = A + nn;                                                              // middle factor
    T *C = B + nn;                                                              // last factor
    T *W = C + nn;                                                              // A*B
    if (GT == 64 && knot >= knots) return;
    const uint32_t k = (uint32_t)(knot % N);
    const size_t blk = (size_t)knot * 3 * nn;

    // Right slot:  R'_k     = -(D_k^-1 R_k) D_{k+1}^-1.
    // Left slot :  L'_k     = -D_k^-1 L_k D_{k-1}^-1, evaluated as the TRANSPOSE of
    //              X = -(D_{k-1}^-1 L_k^T) D_k^-1  -- the same operation sequence the right slot of
    //              knot k-1 runs on (D_{k-1}^-1, R_{k-1}, D_k^-1).  With symmetric D^-1 blocks the two are
    //              equal as matrices for any S, and bit for bit each other's transpose when L_k == R_{k-1}^T.
    for (int side = 0; side < 2; ++side) {  // 0: left slot (needs k-1), 1: right slot (needs k+1)
        if ((side == 0 && k == 0) || (side == 1 && k == N - 1)) continue;  // uniform over the group
        const size_t nb = side == 0 ? blk - (size_t)3 * nn : blk + (size_t)3 * nn;
        for (uint32_t i = tid; i < nn; i += GT) {
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
        group_sync<GT>();
        for (uint32_t i = tid; i < nn; i += GT) {
            const uint32_t c = i / n, r = i - c * n;
            T acc = T(0);
            for (uint32_t q = 0; q < n; ++q) acc = fma_t(A[q * n + r], B[c * n + q], acc);
            W[i] = acc;
        }
        group_sync<GT>();
        for (uint32_t i = tid; i < nn; i += GT) {
            const uint32_t c = i / n, r = i - c * n;
            T acc = T(0);
            for (uint32_t q = 0; q < n; ++q) acc = fma_t(W[q * n + r], C[c * n + q], acc);
            if (side == 1) Pinv[blk + 2 * (size_t)nn + i] = -acc;                  // R'_k(r,c)
            else Pinv[blk + (size_t)r * n + c] = -acc;                             // L'_k(c,r) = X(r,c)
        }
        group_sync<GT>();
    }
}

// ---- compile-time block size, n <= 32: the [D | I] tableau lives in REGISTERS, one column per lane ----
// Lane c < 2n owns tableau column c (n values).  Pivot step j needs column j (held by lane j) in every
// lane: n v_readlane broadcasts into scalars, then n FMAs per lane -- no LDS, no barrier, the pivot loop
// fully unrolled so every register index is static.  Same arithmetic, element for element, as the LDS
// kernel above.  One wavefront per knot, four knots per workgroup.
__device__ __forceinline__ uint32_t pinv_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint64_t pinv_bits(double v) { return __builtin_bit_cast(uint64_t, v); }
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_bcast(double v, int lane)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_reg_kernel(uint32_t N, uint64_t knots, const T *__restrict__ S,
                                                                    T *__restrict__ Pinv, int kind)
{
    constexpr uint32_t n = NCT, nn = n * n;
    __shared__ T stage_all[4][nn];  // D_k^-1 of each wave's knot, for the mirrored write-out
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t knot = (uint64_t)blockIdx.x * 4 + wave;
    if (knot >= knots) return;
    T *stage = stage_all[wave];
    const size_t blk = (size_t)knot * 3 * nn;
    const T *D = S + blk + nn;
    T *out = Pinv + blk;

    T col[n];
#pragma unroll
    for (uint32_t r = 0; r < n; ++r) {
        if (lane < n) col[r] = (kind == 0) ? (r == lane ? T(1) : T(0)) : D[lane * n + r];
        else col[r] = (lane - n == r) ? T(1) : T(0);
    }
    if (kind != 0) {
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            T cj[n];
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) cj[r] = lane_bcast(col[r], (int)j);
            const T piv = T(1) / cj[j];
            const T pr = col[j] * piv;  // scaled pivot-row entry of this lane's column
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) col[r] = (r == j) ? pr : fma_t(-cj[r], pr, col[r]);
        }
    }
    // lanes n .. 2n-1 hold the columns of D_k^-1; mirror the upper triangle (see the LDS kernel) on the way out
    if (lane >= n && lane < 2 * n) {
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) stage[(lane - n) * n + r] = col[r];
    }
    group_sync<64>();
    for (uint32_t i = lane; i < nn; i += 64) {
        const uint32_t c = i / n, r = i - c * n;
        out[nn + i] = r <= c ? stage[c * n + r] : stage[r * n + c];
        // the stair pass overwrites every off-diagonal slot except the two never-read corner blocks
        const uint32_t kk = (uint32_t)(knot % N);
        if (kind != 2 || kk == 0) out[i] = T(0);
        if (kind != 2 || kk == N - 1) out[2 * nn + i] = T(0);
    }
}

// 2n <= 32: TWO knots per wavefront, one in each 32-lane half (lane c < 2n of a half owns tableau column c), so
// 28 + 28 of 64 lanes work instead of 28.  A pivot step needs column j of the half's own knot in every lane
// of that half, which v_readlane cannot give (one scalar per wave): lane j parks its column in LDS and the
// half reads it back as broadcast 16-byte reads -- 8 LDS instructions per step instead of 14 readlanes per
// knot, and a third of the VALU instructions per knot.  Same arithmetic, element for element.
template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_pair_kernel(uint32_t N, uint64_t knots, const T *__restrict__ S,
                                                                     T *__restrict__ Pinv, int kind)
{
    constexpr uint32_t n = NCT, nn = n * n, NP = (n + 3) / 4 * 4;  // column padded to whole 16-byte pieces (fp32)
    static_assert(2 * n <= 32, "two knots per wave need 2n lanes per half");
    __shared__ __attribute__((aligned(16))) T stage_all[4][2][nn];  // D_k^-1 of each half's knot, for the mirrored write-out
    __shared__ __attribute__((aligned(16))) T bcast_all[4][2][NP];  // the pivot column of the current step
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, half = lane >> 5, l = lane & 31u;
    const uint64_t knot = ((uint64_t)blockIdx.x * 4 + wave) * 2 + half;
    const bool alive = knot < knots;
    T *stage = stage_all[wave][half];
    T *bc = bcast_all[wave][half];
    const size_t blk = (size_t)(alive ? knot : 0) * 3 * nn;
    const T *D = S + blk + nn;
    T *out = Pinv + blk;

    T col[n];
#pragma unroll
    for (uint32_t r = 0; r < n; ++r) {
        if (l < n) col[r] = (kind == 0 || !alive) ? (r == l ? T(1) : T(0)) : D[l * n + r];
        else col[r] = (l - n == r) ? T(1) : T(0);
    }
    if (kind != 0) {
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            if (l == j) {
#pragma unroll
                for (uint32_t r = 0; r < n; ++r) bc[r] = col[r];
            }
            group_sync<64>();
            T cj[n];
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) cj[r] = bc[r];
            group_sync<64>();  // everyone has the column before step j+1 overwrites it
            const T piv = T(1) / cj[j];
            const T pr = col[j] * piv;  // scaled pivot-row entry of this lane's column
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) col[r] = (r == j) ? pr : fma_t(-cj[r], pr, col[r]);
        }
    }
    // lanes n .. 2n-1 of a half hold the columns of D_k^-1; mirror the upper triangle on the way out
    if (l >= n && l < 2 * n) {
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) stage[(l - n) * n + r] = col[r];
    }
    group_sync<64>();
    if (alive) {
        const uint32_t kk = (uint32_t)(knot % N);
        for (uint32_t i = l; i < nn; i += 32) {
            const uint32_t c = i / n, r = i - c * n;
            out[nn + i] = r <= c ? stage[c * n + r] : stage[r * n + c];
            // the stair pass overwrites every off-diagonal slot except the two never-read corner blocks
            if (kind != 2 || kk == 0) out[i] = T(0);
            if (kind != 2 || kk == N - 1) out[2 * nn + i] = T(0);
        }
    }
}

// The value lane J of every 16-lane row holds, in all lanes of that row (DPP row_newbcast: a VALU move, no LDS round trip), and the
// in-place Gauss-Jordan elimination of an M x M block held one column per lane on top of it -- the arithmetic of
// pinv_diag_quad_kernel, element for element (the broadcast values are the same numbers that kernel passes through LDS).
#if GBDPCG_PINV_DPP
template <int J> __device__ __forceinline__ float stair_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, true));
}
template <int J> __device__ __forceinline__ double stair_bcast(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + J, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + J, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int J, int M, typename T> __device__ __forceinline__ void stair_eliminate(T (&col)[M], uint32_t l)
{
    if constexpr (J < M) {
        T cj[M];
#pragma unroll
        for (int r = 0; r < M; ++r) cj[r] = stair_bcast<J>(col[r]);
        const T piv = T(1) / cj[J];
        const bool is_j = l == (uint32_t)J;
        const T pr = is_j ? piv : col[J] * piv;
#if GBDPCG_PINV_PACKED
        if constexpr (sizeof(T) == 4 && M % 2 == 0) {
            // two rows per instruction (schur.hip, quad_pivot): the pivot lane's "start from zero" is an exact packed multiply by
            // 0 or 1 instead of a select per row, the update a packed fma -- the same fma on the same numbers, element for element
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
            col[J] = pr;
        } else
#endif
        {
#pragma unroll
            for (int r = 0; r < M; ++r) col[r] = (r == J) ? pr : fma_t(-cj[r], pr, is_j ? T(0) : col[r]);
        }
        stair_eliminate<J + 1, M>(col, l);
    }
}
#endif

// n <= 16: FOUR knots per wavefront, one per 16-lane quarter, with the IN-PLACE form of the same
// elimination: lane c < n of a quarter owns column c of the n x n block only.  The identity half of the
// [D | I] tableau is never stored: its column j stays the unit vector e_j until pivot step j (its pivot-row
// entries are zero up to then) and is created in that step in the place of column j of D, whose pivot-step
// update would only produce e_j.  So lane j computes (r == j ? piv : fma(-cj[r], piv, 0)) -- exactly what
// the tableau lane n+j computes from e_j -- and every other lane the usual update: the same floating-point
// operations on the same numbers as the tableau kernels, with 56 of 64 lanes busy instead of 28.
template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_quad_kernel(uint32_t N, uint64_t knots, const T *__restrict__ S,
                                                                     T *__restrict__ Pinv, int kind)
{
    constexpr uint32_t n = NCT, nn = n * n, NP = (n + 3) / 4 * 4;
    static_assert(n <= 16, "four knots per wave need n lanes per quarter");
    __shared__ __attribute__((aligned(16))) T stage_all[4][4][nn];  // D_k^-1 of each quarter's knot, for the mirrored write-out
    __shared__ __attribute__((aligned(16))) T bcast_all[4][4][NP];  // the pivot column of the current step
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, quarter = lane >> 4, l = lane & 15u;
    const uint64_t knot = ((uint64_t)blockIdx.x * 4 + wave) * 4 + quarter;
    const bool alive = knot < knots;
    T *stage = stage_all[wave][quarter];
    T *bc = bcast_all[wave][quarter];
    const size_t blk = (size_t)(alive ? knot : 0) * 3 * nn;
    const T *D = S + blk + nn;
    const bool owner = l < n;

    T col[n];
#pragma unroll
    for (uint32_t r = 0; r < n; ++r) col[r] = (kind == 0 || !alive || !owner) ? (r == l ? T(1) : T(0)) : D[l * n + r];
#if GBDPCG_PINV_DPP
    (void)bc;
    if (kind != 0) stair_eliminate<0, (int)n>(col, l);
    if (false) {
#else
    if (kind != 0) {
#endif
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            if (l == j) {
#pragma unroll
                for (uint32_t r = 0; r < n; ++r) bc[r] = col[r];
            }
            group_sync<64>();
            T cj[n];
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) cj[r] = bc[r];
            group_sync<64>();  // everyone has the column before step j+1 overwrites it
            const T piv = T(1) / cj[j];
            const bool is_j = l == j;
            const T pr = is_j ? piv : col[j] * piv;  // scaled pivot-row entry of this lane's column
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) col[r] = (r == j) ? pr : fma_t(-cj[r], pr, is_j ? T(0) : col[r]);
        }
    }
    if (owner) {
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) stage[l * n + r] = col[r];
    }
    group_sync<64>();
    // write-out by the whole wave: its four knots are consecutive in memory, so the D slots (and, for the
    // identity / block-Jacobi kinds, the zeroed L and R slots) go out as dense 256-byte stores
    const uint64_t knot0 = ((uint64_t)blockIdx.x * 4 + wave) * 4;
    T *out0 = Pinv + (size_t)knot0 * 3 * nn;
    for (uint32_t e = lane; e < 4 * nn; e += 64) {
        const uint32_t q = e / nn, i = e - q * nn;
        if (knot0 + q >= knots) break;
        const uint32_t c = i / n, r = i - c * n;
        const T *st = stage_all[wave][q];
        T *o = out0 + (size_t)q * 3 * nn;
        o[nn + i] = r <= c ? st[c * n + r] : st[r * n + c];  // mirror the upper triangle: exactly symmetric
        // the stair pass overwrites every off-diagonal slot except the two never-read corner blocks
        const uint32_t kk = (uint32_t)((knot0 + q) % N);
        if (kind != 2 || kk == 0) o[i] = T(0);
        if (kind != 2 || kk == N - 1) o[2 * nn + i] = T(0);
    }
}

// 32 < n <= 64 (BASELINE config 4: n = 36, fp64): ONE knot per wavefront, lane c < n owns column c, the same
// in-place elimination as pinv_diag_quad_kernel (the [D | I] tableau kernels need 2n <= 64 lanes, and the
// LDS form that used to take these sizes spends a workgroup barrier pair per pivot step: 107 us for the 256
// knots of config 4 against a few us here).  The pivot column is broadcast with v_readlane (n scalars per
// step, no LDS, no barrier); the finished inverse is staged in LDS for the mirrored, dense write-out.
template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_diag_wide_kernel(uint32_t N, uint64_t knots, const T *__restrict__ S,
                                                                     T *__restrict__ Pinv, int kind)
{
    constexpr uint32_t n = NCT, nn = n * n;
    static_assert(n > 32 && n <= 64, "one column per lane");
    __shared__ __attribute__((aligned(16))) T stage_all[4][nn];
    const uint32_t wave = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const uint64_t knot = (uint64_t)blockIdx.x * 4 + wave;
    const bool alive = knot < knots;  // wave-uniform; dead waves still run the (cheap, branch-free) loop
    T *stage = stage_all[wave];
    const size_t blk = (size_t)(alive ? knot : 0) * 3 * nn;
    const T *D = S + blk + nn;
    const bool owner = l < n;

    T col[n];
#pragma unroll
    for (uint32_t r = 0; r < n; ++r) col[r] = (kind == 0 || !alive || !owner) ? (r == l ? T(1) : T(0)) : D[(owner ? l : 0u) * n + r];
    if (kind != 0) {
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {  // fully unrolled: every register index is static
            T cj[n];                        // column j, broadcast from lane j into scalars
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) cj[r] = lane_bcast(col[r], (int)j);
            const T piv = T(1) / cj[j];
            const bool is_j = l == j;
            const T pr = is_j ? piv : col[j] * piv;  // scaled pivot-row entry of this lane's column
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) col[r] = (r == j) ? pr : fma_t(-cj[r], pr, is_j ? T(0) : col[r]);
        }
    }
    if (owner) {
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) stage[l * n + r] = col[r];
    }
    group_sync<64>();
    if (!alive) return;
    T *o = Pinv + (size_t)knot * 3 * nn;
    const uint32_t kk = (uint32_t)(knot % N);
    for (uint32_t i = l; i < nn; i += 64) {
        const uint32_t c = i / n, r = i - c * n;
        o[nn + i] = r <= c ? stage[c * n + r] : stage[r * n + c];  // mirror the upper triangle: exactly symmetric
        // the stair pass overwrites every off-diagonal slot except the two never-read corner blocks
        if (kind != 2 || kk == 0) o[i] = T(0);
        if (kind != 2 || kk == N - 1) o[2 * nn + i] = T(0);
    }
}

// Stair off-diagonal slots for compile-time n: one wavefront per knot PAIR (k, k+1), every inner product
// reads BOTH operands as contiguous pairs from LDS.  The wave of knot k produces the right slot of k,
// R'_k = -D_k^-1 R_k D_{k+1}^-1, and the left slot of k+1, L'_{k+1} = -D_{k+1}^-1 L_{k+1} D_k^-1, evaluated as
// the transpose of the same operation sequence with L_{k+1}^T in the place of R_k.  When S is symmetric in
// storage (L_{k+1} == R_k^T bit for bit, tested here by the wave) the second evaluation would repeat the
// first one operation for operation, so its result is written as the mirror image of the first: half the
// work, and Pinv comes out exactly symmetric whenever S is.  D^-1 blocks are exactly symmetric (mirrored by
// pass 1), so row r of the first factor is its column r; the intermediate W = A*B is stored transposed for the
// same reason.  Operation order is the LDS kernel's ((A*B)*C, q ascending), so both produce the same bits.
template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_reg_kernel(uint32_t N, uint64_t knots, const T *__restrict__ S,
                                                                     T *Pinv)
{
    constexpr uint32_t n = NCT, nn = n * n;
    using P2 = typename VecOf<T, 2>::type;
    __shared__ __attribute__((aligned(16))) T lds[4][4 * nn];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t knot = (uint64_t)blockIdx.x * 4 + wave;
    if (knot >= knots) return;
    T *A = lds[wave], *B = A + nn, *C = B + nn, *Wt = C + nn;
    const uint32_t k = (uint32_t)(knot % N);
    if (k == N - 1) return;  // the last knot of a problem has no right neighbour (wave-uniform)
    const size_t blk = (size_t)knot * 3 * nn, nb = blk + (size_t)3 * nn;

    constexpr uint32_t EPL = (nn + 63) / 64;  // elements per lane
    T lt[EPL];                                // L_{k+1}^T, element for element against R_k
    bool differs = false;
#pragma unroll
    for (uint32_t j = 0; j < EPL; ++j) {
        const uint32_t i = lane + 64 * j;
        if (i < nn) {
            const uint32_t c = i / n, r = i - c * n;
            const T rk = S[blk + 2 * (size_t)nn + i];  // R_k(r,c)
            lt[j] = S[nb + (size_t)r * n + c];         // L_{k+1}(c,r)
            differs |= pinv_bits(rk) != pinv_bits(lt[j]);
            A[i] = Pinv[blk + nn + i];                 // D_k^-1
            B[i] = rk;
            C[i] = Pinv[nb + nn + i];                  // D_{k+1}^-1
        }
    }
    const bool symmetric = __builtin_amdgcn_ballot_w64(differs) == 0;  // wave-uniform
    group_sync<64>();
    for (int pass = 0; pass < (symmetric ? 1 : 2); ++pass) {
        if (pass == 1) {  // general S: the left slot of k+1 from its own data
#pragma unroll
            for (uint32_t j = 0; j < EPL; ++j) {
                const uint32_t i = lane + 64 * j;
                if (i < nn) B[i] = lt[j];
            }
            group_sync<64>();
        }
        if constexpr (n % 2 == 0 && (n / 2) * (n / 2) <= 64) {
            // 2 x 2 output tiles, one per lane ((n/2)^2 lanes): two rows of the first factor and two columns of
            // the second feed four accumulators, half the LDS reads per FMA of the one-element form below
            // (which is bound by LDS bandwidth); every output still sums q ascending, so the bits are the same.
            constexpr uint32_t H = n / 2;
            const uint32_t tr = lane / H, tc = lane - tr * H, r0 = 2 * tr, c0 = 2 * tc;
            const bool tile = lane < H * H;
            T w00 = T(0), w01 = T(0), w10 = T(0), w11 = T(0);
            if (tile) {
                const P2 *a0 = reinterpret_cast<const P2 *>(A + r0 * n), *a1 = reinterpret_cast<const P2 *>(A + (r0 + 1) * n);
                const P2 *b0 = reinterpret_cast<const P2 *>(B + c0 * n), *b1 = reinterpret_cast<const P2 *>(B + (c0 + 1) * n);
#pragma unroll
                for (uint32_t q = 0; q < H; ++q) {
                    const P2 x0 = a0[q], x1 = a1[q], y0 = b0[q], y1 = b1[q];
                    w00 = fma_t(x0.x, y0.x, w00); w00 = fma_t(x0.y, y0.y, w00);
                    w01 = fma_t(x0.x, y1.x, w01); w01 = fma_t(x0.y, y1.y, w01);
                    w10 = fma_t(x1.x, y0.x, w10); w10 = fma_t(x1.y, y0.y, w10);
                    w11 = fma_t(x1.x, y1.x, w11); w11 = fma_t(x1.y, y1.y, w11);
                }
                Wt[r0 * n + c0] = w00; Wt[r0 * n + c0 + 1] = w01;            // transposed: row r of W contiguous
                Wt[(r0 + 1) * n + c0] = w10; Wt[(r0 + 1) * n + c0 + 1] = w11;
            }
            group_sync<64>();
            if (tile) {
                const P2 *a0 = reinterpret_cast<const P2 *>(Wt + r0 * n), *a1 = reinterpret_cast<const P2 *>(Wt + (r0 + 1) * n);
                const P2 *b0 = reinterpret_cast<const P2 *>(C + c0 * n), *b1 = reinterpret_cast<const P2 *>(C + (c0 + 1) * n);
                T x00 = T(0), x01 = T(0), x10 = T(0), x11 = T(0);
#pragma unroll
                for (uint32_t q = 0; q < H; ++q) {
                    const P2 u0 = a0[q], u1 = a1[q], y0 = b0[q], y1 = b1[q];
                    x00 = fma_t(u0.x, y0.x, x00); x00 = fma_t(u0.y, y0.y, x00);
                    x01 = fma_t(u0.x, y1.x, x01); x01 = fma_t(u0.y, y1.y, x01);
                    x10 = fma_t(u1.x, y0.x, x10); x10 = fma_t(u1.y, y0.y, x10);
                    x11 = fma_t(u1.x, y1.x, x11); x11 = fma_t(u1.y, y1.y, x11);
                }
                if (pass == 0) {  // R'_k(r,c) at column-major c*n + r
                    T *Rp = Pinv + blk + 2 * (size_t)nn;
                    Rp[c0 * n + r0] = -x00; Rp[c0 * n + r0 + 1] = -x10;
                    Rp[(c0 + 1) * n + r0] = -x01; Rp[(c0 + 1) * n + r0 + 1] = -x11;
                }
                if (pass == 1 || symmetric) {  // L'_{k+1}(c,r) = X(r,c) at column-major r*n + c
                    T *Lp = Pinv + nb;
                    Lp[r0 * n + c0] = -x00; Lp[r0 * n + c0 + 1] = -x01;
                    Lp[(r0 + 1) * n + c0] = -x10; Lp[(r0 + 1) * n + c0 + 1] = -x11;
                }
            }
        } else {
            for (uint32_t i = lane; i < nn; i += 64) {
                const uint32_t c = i / n, r = i - c * n;
                // W(r,c) = sum_q A(r,q) B(q,c);  A symmetric: A(r,q) = A(q,r) = A[r*n + q]
                T acc = T(0);
                if constexpr (n % 2 == 0) {
                    const P2 *ar = reinterpret_cast<const P2 *>(A + r * n), *bc = reinterpret_cast<const P2 *>(B + c * n);
#pragma unroll
                    for (uint32_t q = 0; q < n / 2; ++q) {
                        const P2 a2 = ar[q], b2 = bc[q];
                        acc = fma_t(a2.x, b2.x, acc);
                        acc = fma_t(a2.y, b2.y, acc);
                    }
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < n; ++q) acc = fma_t(A[r * n + q], B[c * n + q], acc);
                }
                Wt[r * n + c] = acc;  // transposed: row r of W contiguous
            }
            group_sync<64>();
            for (uint32_t i = lane; i < nn; i += 64) {
                const uint32_t c = i / n, r = i - c * n;
                T acc = T(0);
                if constexpr (n % 2 == 0) {
                    const P2 *wr = reinterpret_cast<const P2 *>(Wt + r * n), *cc = reinterpret_cast<const P2 *>(C + c * n);
#pragma unroll
                    for (uint32_t q = 0; q < n / 2; ++q) {
                        const P2 w2 = wr[q], c2 = cc[q];
                        acc = fma_t(w2.x, c2.x, acc);
                        acc = fma_t(w2.y, c2.y, acc);
                    }
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < n; ++q) acc = fma_t(Wt[r * n + q], C[c * n + q], acc);
                }
                if (pass == 0) Pinv[blk + 2 * (size_t)nn + i] = -acc;                 // R'_k(r,c)
                if (pass == 1 || symmetric) Pinv[nb + (size_t)r * n + c] = -acc;      // L'_{k+1}(c,r) = X(r,c)
            }
        }
        group_sync<64>();
    }
}

// STAIR in ONE launch for n <= 16 (even n): a workgroup inverts the diagonal blocks of 16 consecutive knots of
// one problem (four per wavefront, pinv_diag_quad_kernel's in-place elimination), keeps the mirrored inverses
// in LDS and evaluates the 15 stair pairs between them from there -- the D^-1 blocks are neither re-read from
// memory nor waited for across a launch boundary.  Consecutive workgroups overlap by one knot (inverted
// twice, written once).  Same operations, same bits as the two-pass form.
// verdicts (optional): one byte per workgroup, 1 when every pair of the chunk had L_{k+1} == R_k^T bit for bit
// in S -- the pairs of Pinv were then written as mirror images, so the same holds for Pinv and a solve that
// follows needs no symmetry test of its own (gbdpcg_form_pinv_solve_*).  Written unconditionally.
// S_SYM: the caller KNOWS L_{k+1} == R_k^T in S (gbdpcg_kkt_step_*): L is never read.  A compile-time switch: as a kernel argument
// the same test cost the ordinary path 40 us of 150 (the compiler's schedule of the pair loop changed).
// (Held to 64 registers -- eight waves per SIMD instead of six, amdgpu_waves_per_eu(8) -- it spills 10 registers and takes 208 us
// instead of 162 for the 1024 x 128 batch: measured in round 3, dropped.)
template <typename T, int NCT, bool S_SYM>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_fused_kernel(uint32_t N, uint32_t chunks, const T *__restrict__ S,
                                                                       T *__restrict__ Pinv, uint8_t *__restrict__ verdicts)
{
    constexpr bool s_symmetric = S_SYM;
    constexpr uint32_t n = NCT, nn = n * n, NP = (n + 3) / 4 * 4, H = n / 2, PAIRS = 15;
    static_assert(n <= 16 && n % 2 == 0, "quarter-wave elimination and 2 x 2 tiles");
    using P2 = typename VecOf<T, 2>::type;
    __shared__ __attribute__((aligned(16))) T inv[16][nn];      // mirrored D^-1 of knots k0 .. k0+15
    __shared__ __attribute__((aligned(16))) T bcast_all[16][NP];
    __shared__ __attribute__((aligned(16))) T work[4][2][nn];   // per wave: B (R_k or L_{k+1}^T) and W^T
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, quarter = lane >> 4, l = lane & 15u;
    const uint32_t prob = blockIdx.x / chunks, chunk = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk * PAIRS;
    const size_t pbase = (size_t)prob * 3 * nn * N;

    {   // ---- pass 1: invert D_{k0 + slot}, slot = 4 wave + quarter
        const uint32_t slot = wave * 4 + quarter, k = k0 + slot;
        const bool alive = k < N, owner = l < n;
        const T *D = S + pbase + (size_t)(alive ? k : 0) * 3 * nn + nn;
        T *bc = bcast_all[slot];
        T col[n];
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) col[r] = (!alive || !owner) ? (r == l ? T(1) : T(0)) : D[l * n + r];
#if GBDPCG_PINV_DPP
        (void)bc;
        stair_eliminate<0, (int)n>(col, l);
#else
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            if (l == j) {
#pragma unroll
                for (uint32_t r = 0; r < n; ++r) bc[r] = col[r];
            }
            group_sync<64>();
            T cj[n];
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) cj[r] = bc[r];
            group_sync<64>();
            const T piv = T(1) / cj[j];
            const bool is_j = l == j;
            const T pr = is_j ? piv : col[j] * piv;
#pragma unroll
            for (uint32_t r = 0; r < n; ++r) col[r] = (r == j) ? pr : fma_t(-cj[r], pr, is_j ? T(0) : col[r]);
        }
#endif
        if (owner) {  // mirrored on the way into LDS: element (r, c) with r > c takes the value of (c, r)
#pragma unroll
            for (uint32_t r = 0; r < n; ++r)
                if (r <= l) inv[slot][l * n + r] = col[r];       // upper triangle of column l, as computed
#pragma unroll
            for (uint32_t r = 0; r < n; ++r)
                if (r < l) inv[slot][r * n + l] = col[r];        // its mirror image: (l, r) := (r, l)
        }
    }
    __syncthreads();

    // ---- D slots: this workgroup owns knots k0 .. k0+14, and N-1 if it is the last.  D'_k leaves with R'_k and L'_{k+1} of its
    // pair below -- the three are one contiguous run of 3 n^2 elements, and memory that sees D' now and the other two blocks of
    // the row ten microseconds later pays for the row twice.  Here: the knot that ends the problem (it starts no pair) and the
    // two corner blocks nobody reads.
    const uint32_t own_end = (chunk == chunks - 1) ? N : min(N, k0 + PAIRS);
    if (own_end == N) {
        T *o = Pinv + pbase + (size_t)(N - 1) * 3 * nn;
        for (uint32_t i = threadIdx.x; i < nn; i += kPinvThreads) {
            o[nn + i] = inv[N - 1 - k0][i];
            o[2 * nn + i] = T(0);
        }
    }
    if (k0 == 0)
        for (uint32_t i = threadIdx.x; i < nn; i += kPinvThreads) Pinv[pbase + i] = T(0);

    // ---- pass 2: pairs (k, k+1), k = k0 + j, j = wave, wave + 4, ...
    T *B = work[wave][0], *Wt = work[wave][1];
    const uint32_t tr = lane / H, tc = lane - tr * H, r0 = 2 * tr, c0 = 2 * tc;
    const bool tile = lane < H * H;
    constexpr uint32_t EPL = (nn + 63) / 64;
    bool any_asymmetric = false;
    for (uint32_t j = wave; j < PAIRS && k0 + j + 1 < N; j += 4) {
        const uint32_t k = k0 + j;
        const T *A = inv[j], *C = inv[j + 1];
        const size_t blk = pbase + (size_t)k * 3 * nn, nb = blk + (size_t)3 * nn;
        // Both blocks are read as they lie in memory (a lane-strided read of L_{k+1}^T costs the address unit one cache line
        // per lane: with the strided R' stores below, that was most of this phase) and L_{k+1} is transposed through LDS.
        T lt[EPL];
        bool differs = false;
#pragma unroll
        for (uint32_t q = 0; q < EPL; ++q) {
            const uint32_t i = lane + 64 * q;
            if (i < nn) {
                B[i] = S[blk + 2 * (size_t)nn + i];   // R_k(r,c) at c n + r
                if (!s_symmetric) Wt[i] = S[nb + i];  // L_{k+1}, as stored (not read at all when the caller knows S symmetric)
            }
        }
        group_sync<64>();
#pragma unroll
        for (uint32_t q = 0; q < EPL; ++q) {
            const uint32_t i = lane + 64 * q;
            lt[q] = T(0);
            if (i < nn && !s_symmetric) {
                const uint32_t c = i / n, r = i - c * n;
                lt[q] = Wt[r * n + c];                // L_{k+1}(c,r)
                differs |= pinv_bits(B[i]) != pinv_bits(lt[q]);
            }
        }
        const bool symmetric = s_symmetric || __builtin_amdgcn_ballot_w64(differs) == 0;  // wave-uniform
        any_asymmetric |= !symmetric;
        group_sync<64>();
        for (int pass = 0; pass < (symmetric ? 1 : 2); ++pass) {
            if (pass == 1) {
#pragma unroll
                for (uint32_t q = 0; q < EPL; ++q) {
                    const uint32_t i = lane + 64 * q;
                    if (i < nn) B[i] = lt[q];
                }
                group_sync<64>();
            }
            if (tile) {
                const P2 *a0 = reinterpret_cast<const P2 *>(A + r0 * n), *a1 = reinterpret_cast<const P2 *>(A + (r0 + 1) * n);
                const P2 *b0 = reinterpret_cast<const P2 *>(B + c0 * n), *b1 = reinterpret_cast<const P2 *>(B + (c0 + 1) * n);
                T w00 = T(0), w01 = T(0), w10 = T(0), w11 = T(0);
#pragma unroll
                for (uint32_t q = 0; q < H; ++q) {
                    const P2 x0 = a0[q], x1 = a1[q], y0 = b0[q], y1 = b1[q];
                    w00 = fma_t(x0.x, y0.x, w00); w00 = fma_t(x0.y, y0.y, w00);
                    w01 = fma_t(x0.x, y1.x, w01); w01 = fma_t(x0.y, y1.y, w01);
                    w10 = fma_t(x1.x, y0.x, w10); w10 = fma_t(x1.y, y0.y, w10);
                    w11 = fma_t(x1.x, y1.x, w11); w11 = fma_t(x1.y, y1.y, w11);
                }
                Wt[r0 * n + c0] = w00; Wt[r0 * n + c0 + 1] = w01;
                Wt[(r0 + 1) * n + c0] = w10; Wt[(r0 + 1) * n + c0 + 1] = w11;
            }
            group_sync<64>();
            if (tile) {
                const P2 *a0 = reinterpret_cast<const P2 *>(Wt + r0 * n), *a1 = reinterpret_cast<const P2 *>(Wt + (r0 + 1) * n);
                const P2 *b0 = reinterpret_cast<const P2 *>(C + c0 * n), *b1 = reinterpret_cast<const P2 *>(C + (c0 + 1) * n);
                T x00 = T(0), x01 = T(0), x10 = T(0), x11 = T(0);
#pragma unroll
                for (uint32_t q = 0; q < H; ++q) {
                    const P2 u0 = a0[q], u1 = a1[q], y0 = b0[q], y1 = b1[q];
                    x00 = fma_t(u0.x, y0.x, x00); x00 = fma_t(u0.y, y0.y, x00);
                    x01 = fma_t(u0.x, y1.x, x01); x01 = fma_t(u0.y, y1.y, x01);
                    x10 = fma_t(u1.x, y0.x, x10); x10 = fma_t(u1.y, y0.y, x10);
                    x11 = fma_t(u1.x, y1.x, x11); x11 = fma_t(u1.y, y1.y, x11);
                }
                // -X' into LDS in the layout of R' (column c0 of X' = row c0 of X ...): R'_k(r,c) = -X(c,r) at c n + r
                Wt[c0 * n + r0] = -x00; Wt[c0 * n + r0 + 1] = -x10;
                Wt[(c0 + 1) * n + r0] = -x01; Wt[(c0 + 1) * n + r0 + 1] = -x11;
            }
            group_sync<64>();
            {   // dense stores: R'_k as it lies in Wt, L'_{k+1}(c,r) = R'_k(r,c) read transposed
                T *Rp = Pinv + blk + 2 * (size_t)nn, *Lp = Pinv + nb;
#pragma unroll
                for (uint32_t q = 0; q < EPL; ++q) {
                    const uint32_t i = lane + 64 * q;
                    if (i < nn) {
                        const uint32_t c = i / n, r = i - c * n;
                        if (pass == 0) {
                            Rp[i - (size_t)nn] = A[i];   // D'_k: the block in front of R'_k
                            Rp[i] = Wt[i];
                        }
                        if (pass == 1 || symmetric) Lp[i] = Wt[r * n + c];
                    }
                }
            }
            group_sync<64>();
        }
    }
    if (verdicts) {  // uniform branch: every wave reaches the barrier
        const int bad = __syncthreads_or(any_asymmetric ? 1 : 0);
        if (threadIdx.x == 0) verdicts[(size_t)prob * chunks + chunk] = bad ? 0 : 1;
    }
}

// (Used for stateSize 16, where the VALU products of the kernel above are the bound: 264 -> 180 us per 1024 x 128 knots.  At 14 and
// 12 the two kernels tie within the box-to-box spread -- neither is bound by its vector work there -- and the older kernel stays.
// GBDPCG_PINV_MFMA_FROM=<n> moves the threshold for A/B runs.)
// The one-launch stair for fp32, rebuilt around what the counters of the kernel above say (profiles/r03_pinv_ab.txt): it ran at
// half the HBM rate with NOTHING saturated -- vector pipes 47-64 % busy, LDS 28 %, 14 GB/s per compute unit -- because every wave
// walks a chain of dependent round trips (D blocks in, eliminate, then four pairs: R / L in, two products, results out) with a
// kilobyte or two in flight.  Two changes:
//   * ONE request per workgroup.  Everything a workgroup reads -- D_k0 .. D_k0+15 and the R_k / L_k+1 blocks between them -- is one
//     contiguous run of 46 n^2 elements of S.  It is pulled into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
//     instruction, no registers) before anything else happens: 36 KB in flight per workgroup, 108 KB per compute unit, and no
//     memory read after it.  Pass 1 and the pair phase read their operands where the DMA put them.
//   * The two n x n x n products of a pair on the MATRIX cores (v_mfma_f32_16x16x4_f32: exact fp32, an fma chain over k --
//     cdna_hip_programming.md, "FP32-input MFMA"): 12 MFMAs per pair (32 cycles of the matrix pipe each, 8 of
//     vector issue) instead of 2 x 56 fma + 56 LDS reads per lane.  The mirrored inverses lie in LDS as zero-padded 16 x 16
//     images (16-float rows): the operand of lane (i, q = lane / 16), A[i][4q .. 4q+3], is ONE ds_read_b128 and k = 4q + kk is
//     the k of MFMA step kk in both operands.  B (R_k, or L_{k+1}^T in the second pass of an asymmetric pair) is read straight
//     from the raw run; its k = 14, 15 "pads" are whatever follows in the run (finite matrix data) and meet the zero pads of
//     the inverse.  T1 = B^T Dk^-1 leaves W = Dk^-1 B in the registers as W[j][4q + i] on lane (j, q): exactly the A operand of
//     X = W Dk1^-1 and the B operand of X^T = Dk1^-1 W^T, so both orientations of the result come out of four more MFMAs
//     each without any lane movement: X^T (lane = row of X) is stored as R'_k, X (lane = column of X) as L'_{k+1} -- 56-byte
//     runs per quarter and register, no transposition through LDS.  Both chains multiply the same pairs of numbers in the same
//     k order (Dk1^-1 is mirrored exactly), so R'_k and L'_{k+1} are bit-for-bit transposes whenever S was: the property the
//     symmetric solve kernels rely on.
// Pass 1 (the inversions) is the DPP Gauss-Jordan of the kernel above; so are the verdict bytes and S_SYM.  S must be 16-byte
// aligned (the launcher checks).  54 KB of LDS per workgroup (n = 14): three workgroups per compute unit.
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef float mf_f32x2 __attribute__((ext_vector_type(2)));

template <int NCT, bool S_SYM>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_mfma_kernel(uint32_t N, uint32_t chunks, const float *__restrict__ S,
                                                                      float *__restrict__ Pinv, uint8_t *__restrict__ verdicts)
{
    constexpr bool s_symmetric = S_SYM;
    constexpr uint32_t n = NCT, nn = n * n, PAIRS = 15, LD = 20, IMG = n * LD;   // images: n rows of 20 floats (16 used: 64-bank-friendly stride)
    static_assert(n <= 16 && n % 2 == 0, "quarter-wave elimination, 16 x 16 MFMA images, 16-byte blocks");
    constexpr uint32_t RAW = 46 * nn;                                  // D_k0 .. D_k0+15 with the R / L blocks between them
    constexpr uint32_t PIECES = RAW / 4, ROUNDS = (PIECES + kPinvThreads - 1) / kPinvThreads;   // 16-byte pieces, per-thread rounds
    __shared__ __attribute__((aligned(16))) float raw[RAW + 8];        // (+8: the k = 14, 15 over-read of the last block)
    __shared__ __attribute__((aligned(16))) float inv[16][IMG];        // mirrored D^-1 of knots k0 .. k0+15, (r, c) at r * 20 + c
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u, quarter = lane >> 4, l = lane & 15u;
    const uint32_t prob = blockIdx.x / chunks, chunk = blockIdx.x - prob * chunks;
    const uint32_t k0 = chunk * PAIRS;
    const size_t pbase = (size_t)prob * 3 * nn * N;

    {   // ---- the workgroup's whole input, one request: pieces past the end of the problem re-read its last piece (never used)
        const size_t g0 = pbase + (size_t)k0 * 3 * nn + nn;            // D_k0
        const size_t left = (size_t)3 * nn * N - ((size_t)k0 * 3 * nn + nn);   // elements from there to the end of the problem
        const uint32_t have = left / 4 < PIECES ? (uint32_t)(left / 4) : PIECES;
        const float *src = S + g0;
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
            const uint32_t piece = rd * kPinvThreads + wave * 64 + lane;
            const uint32_t off = (piece < have ? piece : have - 1u) * 16u;
            const uint32_t dst = (uint32_t)(uintptr_t)raw + (rd * kPinvThreads + wave * 64) * 16u;   // wave-uniform; the lane's 16 bytes follow
            if (piece < PIECES) {   // (lanes past the run are masked off: an inactive lane of an LDS-DMA writes nothing)
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(src), "v"(off), "s"(dst) : "memory");
            }
        }
        if (threadIdx.x < 8) raw[RAW + threadIdx.x] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    {   // ---- pass 1: invert D_{k0 + slot}, slot = 4 wave + quarter
        const uint32_t slot = wave * 4 + quarter, k = k0 + slot;
        const bool alive = k < N, owner = l < n;
        const float *D = raw + slot * 3 * nn;
        float col[n];
#pragma unroll
        for (uint32_t r = 0; r < n; ++r) col[r] = (!alive || !owner) ? (r == l ? 1.f : 0.f) : D[l * n + r];
#if !(GBDPCG_PINV_SKIP & 1)
        stair_eliminate<0, (int)n>(col, l);
#endif
        if (owner) {  // mirrored on the way into LDS: element (r, c) with r > c takes the value of (c, r)
#pragma unroll
            for (uint32_t r = 0; r < n; ++r)
                if (r <= l) inv[slot][l * LD + r] = col[r];       // upper triangle of column l, as computed
#pragma unroll
            for (uint32_t r = 0; r < n; ++r)
                if (r < l) inv[slot][r * LD + l] = col[r];        // its mirror image: (l, r) := (r, l)
        }
        // zero pads: columns n .. 15 of the image (what they meet in the other operand is finite, so the sum is exact)
        if constexpr (n < 16) {
            if (owner) {
#pragma unroll
                for (uint32_t c = n; c < 16; ++c) inv[slot][l * LD + c] = 0.f;
            }
        }
    }
    __syncthreads();

    // ---- D slots of the knot that ends the problem and the two corner blocks nobody reads (see the kernel above)
    const uint32_t own_end = (chunk == chunks - 1) ? N : min(N, k0 + PAIRS);
    if (own_end == N) {
        float *o = Pinv + pbase + (size_t)(N - 1) * 3 * nn;
        for (uint32_t i = threadIdx.x; i < nn; i += kPinvThreads) {
            const uint32_t c = i / n, r = i - c * n;
            o[nn + i] = inv[N - 1 - k0][c * LD + r];
            o[2 * nn + i] = 0.f;
        }
    }
    if (k0 == 0)
        for (uint32_t i = threadIdx.x; i < nn; i += kPinvThreads) Pinv[pbase + i] = 0.f;

    // ---- pass 2: pairs (k, k+1), k = k0 + j, j = wave, wave + 4, ...: nothing but LDS reads, MFMAs and stores
    constexpr uint32_t EPL = (nn + 63) / 64;
    const uint32_t lc = l < n ? l : n - 1;   // lanes 14, 15 of a quarter compute rows / columns nobody stores: any finite operand
    bool any_asymmetric = false;
    for (uint32_t j = wave; j < PAIRS && k0 + j + 1 < N; j += 4) {
        const uint32_t k = k0 + j;
        const float *A = inv[j], *C = inv[j + 1];
        const float *Rk = raw + j * 3 * nn + nn, *Lk1 = Rk + nn;          // R_k(r,c) at c n + r, L_{k+1}(r,c) likewise
        const size_t blk = pbase + (size_t)k * 3 * nn, nb = blk + (size_t)3 * nn;
        bool differs = false;
        if (!s_symmetric) {
#pragma unroll
            for (uint32_t q = 0; q < EPL; ++q) {
                const uint32_t i = lane + 64 * q;
                if (i < nn) {
                    const uint32_t c = i / n, r = i - c * n;
                    differs |= pinv_bits(Rk[i]) != pinv_bits(Lk1[r * n + c]);   // R_k(r,c) against L_{k+1}(c,r)
                }
            }
        }
        const bool symmetric = s_symmetric || __builtin_amdgcn_ballot_w64(differs) == 0;  // wave-uniform
        any_asymmetric |= !symmetric;
        // the D'_k block goes out in front of R'_k (one contiguous run of the row)
        {
            float *Dp = Pinv + blk + (size_t)nn;
#pragma unroll
            for (uint32_t q = 0; q < EPL; ++q) {
                const uint32_t i = lane + 64 * q;
                if (i < nn) {
                    const uint32_t c = i / n, r = i - c * n;
#if GBDPCG_PINV_SKIP & 4
                    if (A[c * LD + r] == 1.2345f)
#endif
                    Dp[i] = A[c * LD + r];
                }
            }
        }
        const mf_f32x4 a4 = *reinterpret_cast<const mf_f32x4 *>(A + lc * LD + 4 * quarter);   // Dk^-1 (4q + kk, l), symmetric
        const mf_f32x4 c4 = *reinterpret_cast<const mf_f32x4 *>(C + lc * LD + 4 * quarter);   // Dk1^-1 (l, 4q + kk) = (4q + kk, l)
        for (int pass = 0; pass < (symmetric ? 1 : 2); ++pass) {
            mf_f32x4 b4;   // B(4q + kk, l): B = R_k, then (asymmetric pair) L_{k+1}^T
            if (pass == 0) {
                const mf_f32x2 lo = *reinterpret_cast<const mf_f32x2 *>(Rk + lc * n + 4 * quarter);
                const mf_f32x2 hi = *reinterpret_cast<const mf_f32x2 *>(Rk + lc * n + 4 * quarter + 2);
                b4 = mf_f32x4{lo.x, lo.y, hi.x, hi.y};
            } else {
                const float *col0 = Lk1 + (4 * quarter) * n + lc;     // L_{k+1}(l, 4q + kk) at (4q + kk) n + l
                b4 = mf_f32x4{col0[0], col0[n], col0[2 * n], col0[3 * n]};
            }
            mf_f32x4 t = {0.f, 0.f, 0.f, 0.f}, x = t, xt = t;
#if GBDPCG_PINV_SKIP & 2
            x = b4 + a4; xt = b4 + c4;
#else
            // T1 = B^T Dk^-1: t[i] on lane (j, q) = W[j][4q + i], W = Dk^-1 B
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.x, a4.x, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.y, a4.y, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.z, a4.z, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(b4.w, a4.w, t, 0, 0, 0);
            // X = W Dk1^-1 (lane = column of X) and X^T = Dk1^-1 W^T (lane = row of X): the same products in the same order
            x = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, c4.x, x, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.x, t.x, xt, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, c4.y, x, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.y, t.y, xt, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, c4.z, x, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.z, t.z, xt, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, c4.w, x, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f32_16x16x4f32(c4.w, t.w, xt, 0, 0, 0);
#endif
            // R'_k(r, c) = -X(r, c) at c n + r: lane (j, q'), register i holds X^T[4q' + i][j] = X(j, 4q' + i)
            // L'_{k+1}(r, c) = -X(c, r) at c n + r: lane (j, q'), register i holds X[4q' + i][j]
            float *Rp = Pinv + blk + 2 * (size_t)nn, *Lp = Pinv + nb;
            const float xr[4] = {xt.x, xt.y, xt.z, xt.w}, xl[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t c = 4 * quarter + i;
#if GBDPCG_PINV_SKIP & 4
                if (l < n && c < n && xr[i] == 1.2345f && xl[i] == 5.4321f) {   // timing build: no result store
#else
                if (l < n && c < n) {
#endif
                    if (pass == 0) Rp[c * n + l] = -xr[i];
                    if (pass == 1 || symmetric) Lp[c * n + l] = -xl[i];
                }
            }
        }
    }
    if (verdicts) {  // uniform branch: every wave reaches the barrier
        const int bad = __syncthreads_or(any_asymmetric ? 1 : 0);
        if (threadIdx.x == 0) verdicts[(size_t)prob * chunks + chunk] = bad ? 0 : 1;
    }
}

// Stair slots for 32 < n <= 64 (even n; BASELINE config 4): one WORKGROUP per knot pair (k, k+1).  The same
// operation sequence as the wave-per-pair kernels above -- W = D_k^-1 R_k with 2 x 2 register tiles, X = W D_{k+1}^-1,
// R'_k = -X, and L'_{k+1} = -X^T written as the mirror image when the workgroup finds L_{k+1} == R_k^T in S (else a
// second pass on L_{k+1}^T) -- with the (n/2)^2 tiles dealt over 256 threads and the factors staged in LDS.  The
// runtime-n kernel this replaces evaluates every pair twice (once from each side), with scalar inner loops.
template <typename T, int NCT>
__global__ __launch_bounds__(kPinvThreads) void pinv_stair_wide_kernel(uint32_t N, const T *__restrict__ S, T *__restrict__ Pinv)
{
    constexpr uint32_t n = NCT, nn = n * n, H = n / 2, TILES = H * H;
    static_assert(n > 32 && n <= 64 && n % 2 == 0 && 4 * nn * sizeof(T) <= 64 * 1024, "four n x n factors in static LDS");
    using P2 = typename VecOf<T, 2>::type;
    __shared__ __attribute__((aligned(16))) T A[nn], B[nn], C[nn], Wt[nn];
    const uint32_t pairs_per_problem = N - 1;
    const uint32_t prob = blockIdx.x / pairs_per_problem, k = blockIdx.x - prob * pairs_per_problem;
    const size_t blk = ((size_t)prob * N + k) * 3 * nn, nb = blk + (size_t)3 * nn;
    constexpr uint32_t EPL = (nn + kPinvThreads - 1) / kPinvThreads;
    T lt[EPL];
    bool differs = false;
#pragma unroll
    for (uint32_t q = 0; q < EPL; ++q) {
        const uint32_t i = threadIdx.x + kPinvThreads * q;
        if (i < nn) {
            const uint32_t c = i / n, r = i - c * n;
            const T rk = S[blk + 2 * (size_t)nn + i];   // R_k(r,c)
            lt[q] = S[nb + (size_t)r * n + c];          // L_{k+1}(c,r)
            differs |= pinv_bits(rk) != pinv_bits(lt[q]);
            B[i] = rk;
            A[i] = Pinv[blk + nn + i];                  // D_k^-1 (mirrored by the diagonal pass: exactly symmetric)
            C[i] = Pinv[nb + nn + i];                   // D_{k+1}^-1
        }
    }
    const bool symmetric = __syncthreads_or(differs ? 1 : 0) == 0;  // also the barrier after the staging stores
    for (int pass = 0; pass < (symmetric ? 1 : 2); ++pass) {
        if (pass == 1) {
#pragma unroll
            for (uint32_t q = 0; q < EPL; ++q) {
                const uint32_t i = threadIdx.x + kPinvThreads * q;
                if (i < nn) B[i] = lt[q];
            }
            __syncthreads();
        }
        for (uint32_t t = threadIdx.x; t < TILES; t += kPinvThreads) {
            const uint32_t tr = t / H, tc = t - tr * H, r0 = 2 * tr, c0 = 2 * tc;
            const P2 *a0 = reinterpret_cast<const P2 *>(A + r0 * n), *a1 = reinterpret_cast<const P2 *>(A + (r0 + 1) * n);
            const P2 *b0 = reinterpret_cast<const P2 *>(B + c0 * n), *b1 = reinterpret_cast<const P2 *>(B + (c0 + 1) * n);
            T w00 = T(0), w01 = T(0), w10 = T(0), w11 = T(0);
#pragma unroll
            for (uint32_t q = 0; q < H; ++q) {
                const P2 x0 = a0[q], x1 = a1[q], y0 = b0[q], y1 = b1[q];
                w00 = fma_t(x0.x, y0.x, w00); w00 = fma_t(x0.y, y0.y, w00);
                w01 = fma_t(x0.x, y1.x, w01); w01 = fma_t(x0.y, y1.y, w01);
                w10 = fma_t(x1.x, y0.x, w10); w10 = fma_t(x1.y, y0.y, w10);
                w11 = fma_t(x1.x, y1.x, w11); w11 = fma_t(x1.y, y1.y, w11);
            }
            Wt[r0 * n + c0] = w00; Wt[r0 * n + c0 + 1] = w01;
            Wt[(r0 + 1) * n + c0] = w10; Wt[(r0 + 1) * n + c0 + 1] = w11;
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < TILES; t += kPinvThreads) {
            const uint32_t tr = t / H, tc = t - tr * H, r0 = 2 * tr, c0 = 2 * tc;
            const P2 *a0 = reinterpret_cast<const P2 *>(Wt + r0 * n), *a1 = reinterpret_cast<const P2 *>(Wt + (r0 + 1) * n);
            const P2 *b0 = reinterpret_cast<const P2 *>(C + c0 * n), *b1 = reinterpret_cast<const P2 *>(C + (c0 + 1) * n);
            T x00 = T(0), x01 = T(0), x10 = T(0), x11 = T(0);
#pragma unroll
            for (uint32_t q = 0; q < H; ++q) {
                const P2 u0 = a0[q], u1 = a1[q], y0 = b0[q], y1 = b1[q];
                x00 = fma_t(u0.x, y0.x, x00); x00 = fma_t(u0.y, y0.y, x00);
                x01 = fma_t(u0.x, y1.x, x01); x01 = fma_t(u0.y, y1.y, x01);
                x10 = fma_t(u1.x, y0.x, x10); x10 = fma_t(u1.y, y0.y, x10);
                x11 = fma_t(u1.x, y1.x, x11); x11 = fma_t(u1.y, y1.y, x11);
            }
            if (pass == 0) {
                T *Rp = Pinv + blk + 2 * (size_t)nn;
                Rp[c0 * n + r0] = -x00; Rp[c0 * n + r0 + 1] = -x10;
                Rp[(c0 + 1) * n + r0] = -x01; Rp[(c0 + 1) * n + r0 + 1] = -x11;
            }
            if (pass == 1 || symmetric) {
                T *Lp = Pinv + nb;
                Lp[r0 * n + c0] = -x00; Lp[r0 * n + c0 + 1] = -x01;
                Lp[(r0 + 1) * n + c0] = -x10; Lp[(r0 + 1) * n + c0 + 1] = -x11;
            }
        }
        __syncthreads();
    }
}

template <typename T, int GT, int EPT_MAX>
static hipError_t launch_form_pinv_g(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *S, T *Pinv,
                                     int kind, hipStream_t s)
{
    constexpr uint32_t GROUPS = kPinvThreads / GT;
    const size_t lds1 = (size_t)GROUPS * align16<T>(2 * n * n) * sizeof(T);
    const size_t lds2 = (size_t)GROUPS * align16<T>(4 * n * n) * sizeof(T);
    if (lds1 > dev.lds_per_wg_max || lds2 > dev.lds_per_wg_max) return hipErrorInvalidValue;
    const uint64_t knots = (uint64_t)N * batch;
    const uint64_t blocks = (knots + GROUPS - 1) / GROUPS;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    auto k1 = pinv_diag_kernel<T, GT, EPT_MAX>;
    auto k2 = pinv_stair_kernel<T, GT>;
    if (lds1 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
    }
    if (lds2 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k1, dim3((uint32_t)blocks), dim3(kPinvThreads), lds1, s, n, N, knots, S, Pinv, kind);
    if (kind == 2) hipLaunchKernelGGL(k2, dim3((uint32_t)blocks), dim3(kPinvThreads), lds2, s, n, N, knots, S, Pinv);
    return hipGetLastError();
}

template <typename T, int GT>
static hipError_t launch_stair_only(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *S, T *Pinv,
                                    hipStream_t s)
{
    constexpr uint32_t GROUPS = kPinvThreads / GT;
    const size_t lds2 = (size_t)GROUPS * align16<T>(4 * n * n) * sizeof(T);
    if (lds2 > dev.lds_per_wg_max) return hipErrorInvalidValue;
    const uint64_t knots = (uint64_t)N * batch, blocks = (knots + GROUPS - 1) / GROUPS;
    auto k2 = pinv_stair_kernel<T, GT>;
    if (lds2 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k2, dim3((uint32_t)blocks), dim3(kPinvThreads), lds2, s, n, N, knots, S, Pinv);
    return hipGetLastError();
}

// Workgroups per problem of the one-launch stair kernel when launch_form_pinv will use it for this shape (it is
// the only form that can report symmetry verdicts), else 0.
template <typename T> uint32_t pinv_verdict_chunks(uint32_t n, uint32_t N, int kind)
{
    static const bool two_pass = getenv("GBDPCG_PINV_TWO_PASS") != nullptr;  // tuning runs only
    if (kind != 2 || N < 2 || two_pass || n > 16 || n % 2) return 0;
    bool specialised = false;
#define GBDPCG_CASE(NN) specialised |= n == NN;
    GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    return specialised ? (N - 1 + 14) / 15 : 0;
}

// Block sizes with compile-time formation kernels: the streaming kernels' list plus the odd sizes and 22, which otherwise take the
// runtime-n LDS kernels (stair of 1024 x 128 at n = 15: 1,093 us, at n = 22: 3,055 us).
#define GBDPCG_PINV_N(X) GBDPCG_SPECIALIZED_N(X) X(3) X(5) X(7) X(9) X(11) X(15) X(22)

template <typename T>
hipError_t launch_form_pinv(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *S, T *Pinv,
                            int kind, hipStream_t s, uint8_t *verdicts, bool s_symmetric)
{
    if (verdicts && pinv_verdict_chunks<T>(n, N, kind) == 0) return hipErrorInvalidValue;  // caller asks first
    // register-resident tableau for the compile-time block sizes with 2n <= 64 lanes
#define GBDPCG_CASE(NN)                                                                                              \
    if constexpr (2 * NN <= 64) {                                                                                    \
        if (n == NN) {                                                                                               \
            const uint64_t knots = (uint64_t)N * batch, blocks = (knots + 3) / 4;                                    \
            if (blocks > 0x7fffffffull) return hipErrorInvalidValue;                                                 \
            if constexpr (NN <= 16 && NN % 2 == 0) {                                                                 \
                const uint32_t chunks = pinv_verdict_chunks<T>(n, N, kind);                                          \
                if (chunks) {                                                                                        \
                    if ((uint64_t)chunks * batch > 0x7fffffffull) return hipErrorInvalidValue;                       \
                    if constexpr (sizeof(T) == 4 && NN >= 12) {   /* measured: -32 % at 16; a tie at 14 and 12 (profiles/r03_pinv_ab.txt) */ \
                        static const bool no_mfma = getenv("GBDPCG_PINV_NO_MFMA") != nullptr; /* tuning runs only */ \
                        static const int mfma_from = [] { const char *e = getenv("GBDPCG_PINV_MFMA_FROM"); return e ? atoi(e) : 16; }(); \
                        if (!no_mfma && (int)NN >= mfma_from && reinterpret_cast<uintptr_t>(S) % 16 == 0) {   /* (LDS-DMA in 16-byte pieces) */ \
                            const dim3 grid(chunks * batch);                                                         \
                            if (s_symmetric)                                                                         \
                                hipLaunchKernelGGL((pinv_stair_mfma_kernel<NN, true>), grid, dim3(kPinvThreads), 0, s, N, chunks, \
                                                   (const float *)S, (float *)Pinv, verdicts);                       \
                            else                                                                                     \
                                hipLaunchKernelGGL((pinv_stair_mfma_kernel<NN, false>), grid, dim3(kPinvThreads), 0, s, N, chunks, \
                                                   (const float *)S, (float *)Pinv, verdicts);                       \
                            return hipGetLastError();                                                                \
                        }                                                                                            \
                    }                                                                                                \
                    if (s_symmetric)                                                                                     \
                        hipLaunchKernelGGL((pinv_stair_fused_kernel<T, NN, true>), dim3(chunks * batch), dim3(kPinvThreads), 0,  \
                                           s, N, chunks, S, Pinv, verdicts);                                             \
                    else                                                                                                 \
                        hipLaunchKernelGGL((pinv_stair_fused_kernel<T, NN, false>), dim3(chunks * batch), dim3(kPinvThreads), 0, \
                                           s, N, chunks, S, Pinv, verdicts);                                                   \
                    return hipGetLastError();                                                                        \
                }                                                                                                    \
            }                                                                                                        \
            if constexpr (2 * NN <= 32) {                                                                            \
                static const bool force_pair = getenv("GBDPCG_PINV_PAIR") != nullptr; /* tuning runs only */        \
                if (NN <= 16 && !force_pair && kind == 2) {                                                          \
                    hipLaunchKernelGGL((pinv_diag_quad_kernel<T, NN>), dim3((uint32_t)((knots + 15) / 16)),         \
                                       dim3(kPinvThreads), 0, s, N, knots, S, Pinv, kind);                           \
                } else {                                                                                             \
                    hipLaunchKernelGGL((pinv_diag_pair_kernel<T, NN>), dim3((uint32_t)((knots + 7) / 8)),           \
                                       dim3(kPinvThreads), 0, s, N, knots, S, Pinv, kind);                           \
                }                                                                                                    \
            } else {                                                                                                 \
                hipLaunchKernelGGL((pinv_diag_reg_kernel<T, NN>), dim3((uint32_t)blocks), dim3(kPinvThreads), 0, s, N, knots, \
                                   S, Pinv, kind);                                                                   \
            }                                                                                                        \
            if (kind != 2) return hipGetLastError();                                                                 \
            hipLaunchKernelGGL((pinv_stair_reg_kernel<T, NN>), dim3((uint32_t)blocks), dim3(kPinvThreads), 0, s, N, knots, S, \
                               Pinv);                                                                                \
            return hipGetLastError();                                                                                \
        }                                                                                                            \
    }
    GBDPCG_PINV_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    // compile-time block sizes with one column per lane (32 < n <= 64): in-place elimination, one knot per wave;
    // the stair slots then come from the runtime-n LDS kernel
#define GBDPCG_CASE(NN)                                                                                              \
    if constexpr (NN > 32 && NN <= 64) {                                                                             \
        static const bool no_wide = getenv("GBDPCG_PINV_NO_WIDE") != nullptr; /* tuning runs only */                 \
        if (n == NN && !no_wide) {                                                                                   \
            const uint64_t knots = (uint64_t)N * batch, blocks = (knots + 3) / 4;                                    \
            if (blocks > 0x7fffffffull) return hipErrorInvalidValue;                                                 \
            hipLaunchKernelGGL((pinv_diag_wide_kernel<T, NN>), dim3((uint32_t)blocks), dim3(kPinvThreads), 0, s, N, knots, S, \
                               Pinv, kind);                                                                          \
            if (kind != 2 || N < 2) return hipGetLastError();                                                        \
            if constexpr (NN % 2 == 0 && 4 * NN * NN * sizeof(T) <= 64 * 1024) {                                     \
                if ((uint64_t)(N - 1) * batch > 0x7fffffffull) return hipErrorInvalidValue;                          \
                hipLaunchKernelGGL((pinv_stair_wide_kernel<T, NN>), dim3((N - 1) * batch), dim3(kPinvThreads), 0, s, N, S, Pinv); \
                return hipGetLastError();                                                                            \
            } else {                                                                                                 \
                return launch_stair_only<T, 256>(dev, n, N, batch, S, Pinv, s);                                     \
            }                                                                                                        \
        }                                                                                                            \
    }
    GBDPCG_PINV_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    // wave-per-knot needs the whole [D|I] tableau to fit 16 elements per lane: 2 n^2 <= 1024
    // (EPT_MAX tableau elements per thread are held in registers across the read / write halves of a pivot step)
    if (2 * n * n <= 16 * 64) return launch_form_pinv_g<T, 64, 16>(dev, n, N, batch, S, Pinv, kind, s);
    if (2 * n * n <= 16 * 256) return launch_form_pinv_g<T, 256, 16>(dev, n, N, batch, S, Pinv, kind, s);
    if (2 * n * n <= 64 * 256) return launch_form_pinv_g<T, 256, 64>(dev, n, N, batch, S, Pinv, kind, s);
    return hipErrorInvalidValue;
}

template hipError_t launch_form_pinv<float>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const float *,
                                            float *, int, hipStream_t, uint8_t *, bool);
template hipError_t launch_form_pinv<double>(const DeviceInfo &, uint32_t, uint32_t, uint32_t, const double *,
                                             double *, int, hipStream_t, uint8_t *, bool);
template uint32_t pinv_verdict_chunks<float>(uint32_t, uint32_t, int);
template uint32_t pinv_verdict_chunks<double>(uint32_t, uint32_t, int);

}  // namespace gbdpcg
