// pcg_persist_kernels.hpp -- (the kernels of pcg_persist.hip, in a header so that their instantiations compile in three units side by side:
// pcg_persist.hip, pcg_persist_b.hip, pcg_persist_c.hip)
//
// PCG for ONE large problem spread over many CUs inside a single persistent launch.
//
// Replaces pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218) for problems whose vectors do not fit one
// workgroup (BASELINE config 4: n = 36, N = 256, fp64).  The reference gives every knot a block, keeps the
// knot's block-rows of S and Pinv in shared memory for the whole solve (pcg.cuh:104-110) and crosses the grid
// four times per iteration with grid.sync() (pcg.cuh:166,178,190,207).  Here:
//
//   * a workgroup owns K consecutive knots (K = 1, 2 or 4; 256 threads per knot) and keeps their block-rows
//     [L|D|R] of S and Pinv in REGISTERS for the whole solve: thread (row, group) holds a contiguous run of
//     COLS columns of its row of both matrices (n = 36: 7 groups x 16 columns, 64 VGPRs in fp64).  The matrices
//     are read from HBM once per solve; an iteration touches only LDS and the hand-off words below.
//   * the iteration is cut at its two inner products, as in the split path, but the cut is an in-kernel
//     ALL-GATHER instead of a kernel boundary: after a product every workgroup publishes, in one go, its
//     partial of the inner product AND the two boundary knots of the product vector that its neighbours need
//     (the residual / direction of a neighbour's halo knot is then updated redundantly, so the two axpy
//     barriers of the reference disappear).  Every workgroup then sums all partials in the same order -> the same
//     bits everywhere -> the exit test of pcg.cuh:195 stays uniform, exactly what pcg.cuh:147,167,191 ensure.
//     Two such round trips per iteration: the inner products of textbook PCG depend on each other.
//   * hand-off words are data-tagged 8-byte granules {epoch, 32 value bits} written and polled with agent-scope
//     relaxed atomics (sc1 write-through stores / sc1 loads: MI355X_MICROARCH.md, "Valid forms", R2): the data is
//     the flag, so there is no fence and no separate flag round trip.  An fp64 value is two granules.  Slots are
//     double-buffered by epoch parity (a producer can be at most one epoch ahead of any consumer); epochs continue
//     from a per-problem base kept in the workspace, so nothing has to be cleared between launches and the whole
//     solve is ONE kernel node in a hipGraph.
//   * one workgroup per CU at most (grid <= CU count), every spin is bounded: a launch that cannot get all its
//     workgroups resident gives up instead of hanging -- nothing of the problem is written -- and the workgroup that
//     finishes LAST (an agent-scope counter tells it; the others have left by then) solves the problem alone,
//     streaming, inside this same launch (pcg_stream.hpp, stream_rescue): no caller sees an unsolved problem.  The
//     same workgroup stores the epoch base of the next launch: a workgroup that only got onto the device after the
//     others had given up still publishes under THIS launch's epochs, and whatever it leaves in its slots is older than
//     anything the next launch polls for.
#pragma once

#include <cstdlib>
#include <type_traits>

#include "pcg_stream.hpp"

namespace gbdpcg {

typedef unsigned long long u64;

#define GBDPCG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// u64 control words per problem (lines of their own): [0] epoch base of the next launch, [1..15] stamps of the diagnostic
// build, [16] workgroups of the running launch that have finished
constexpr uint32_t kPersistCtrl = 32, kPersistFinished = 16;

// Leaving a problem's launch (all threads of the workgroup).  The last workgroup to get here stores the next launch's epoch
// base -- every workgroup has read `base` by then, and every granule this launch wrote carries a tag below base + span --
// and, if any workgroup reports that the launch gave up, solves the problem alone (the others have left and have written
// nothing).  [kPersistFinished] counts the workgroups that have left, [kPersistFinished + 1] is non-zero once one of them
// gave up; the last one puts both back.  red: 2 * WAVES elements of LDS, flag: one word of LDS.
template <typename T, int WAVES>
__device__ __forceinline__ void persist_leave(const PcgArgs<T> &a, uint32_t prob, u64 *ws, uint32_t W, uint32_t base, uint32_t span,
                                              bool failed, T *red, uint32_t *flag)
{
    if (threadIdx.x == 0) {
        if (failed) __hip_atomic_fetch_max(ws + kPersistFinished + 1, 1ull, GBDPCG_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u64 before = __hip_atomic_fetch_add(ws + kPersistFinished, 1ull, GBDPCG_RLX_AGENT);
        uint32_t rescue = 0u;
        if (before == W - 1u) {
            rescue = __hip_atomic_exchange(ws + kPersistFinished + 1, 0ull, GBDPCG_RLX_AGENT) != 0ull ? 1u : 0u;
            __hip_atomic_store(ws + kPersistFinished, 0ull, GBDPCG_RLX_AGENT);
            __hip_atomic_store(ws, (u64)(base + span), GBDPCG_RLX_AGENT);
        }
        *flag = rescue;
    }
    __syncthreads();
    if (*flag != 0u && !a.rescue_off && a.rescue_vec)
        stream_rescue<T, WAVES>(a, prob, reinterpret_cast<T *>(a.rescue_vec) + (size_t)prob * rescue_vec_elems<T>(a.n, a.N), red);
}

// tests/test_gpu_persist.py, variants/libgbdpcg_hooks.so only (-DGBDPCG_TEST_HOOKS): hold workgroup `hold_wg` of every
// problem back for `hold_us` microseconds before it reads anything -- a workgroup that gets onto the device late.
#ifdef GBDPCG_TEST_HOOKS
#define GBDPCG_PERSIST_HOLD(BLOCK, HOLD_WG, HOLD_US)                                              \
    if ((HOLD_US) != 0u && (BLOCK) == (HOLD_WG)) {                                                \
        const u64 until = __builtin_amdgcn_s_memrealtime() + ((HOLD_US) == 0xffffffffu ? 0ull : (u64)(HOLD_US) * 100ull); \
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(64);            \
    }
// ... and GBDPCG_PERSIST_DROP_WG: the last workgroup of the grid leaves at once, publishing nothing (hold_us == 0xffffffff)
#define GBDPCG_PERSIST_DROP(WS, W, BASE, SPAN, HOLD_US)                                           \
    if ((HOLD_US) == 0xffffffffu && blockIdx.x == gridDim.x - 1u) {                               \
        persist_leave<T, NWAVES>(a, prob, WS, W, BASE, SPAN, true, rescue_red, bci + 3);          \
        return;                                                                                   \
    }
#else
#define GBDPCG_PERSIST_HOLD(BLOCK, HOLD_WG, HOLD_US)
#define GBDPCG_PERSIST_DROP(WS, W, BASE, SPAN, HOLD_US)
#endif

template <typename T> struct Gran;
template <> struct Gran<float> { static constexpr uint32_t PER = 1; };
template <> struct Gran<double> { static constexpr uint32_t PER = 2; };

__device__ __forceinline__ void gran_store(u64 *slot, uint32_t epoch, float v)
{
    __hip_atomic_store(slot, ((u64)epoch << 32) | __builtin_bit_cast(uint32_t, v), GBDPCG_RLX_AGENT);
}
__device__ __forceinline__ void gran_store(u64 *slot, uint32_t epoch, double v)
{
    const u64 b = __builtin_bit_cast(u64, v);
    __hip_atomic_store(slot, ((u64)epoch << 32) | (b & 0xffffffffull), GBDPCG_RLX_AGENT);
    __hip_atomic_store(slot + 1, ((u64)epoch << 32) | (b >> 32), GBDPCG_RLX_AGENT);
}
// One value slot = Gran<T>::PER adjacent granules; read with ONE sc1 buffer load (8 bytes for fp32, 16 for fp64: each
// 8-byte half carries its own tag, so a torn 16-byte read is detected, not consumed).  off = byte offset in the region.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;   // cache-policy bits of the raw buffer builtins on gfx94x/gfx950: bit 4 = sc1

__device__ __forceinline__ bool slot_load(__amdgpu_buffer_rsrc_t region, uint32_t off, uint32_t epoch, float &v)
{
    const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(region, (int)off, 0, kAuxSc1);
    v = __builtin_bit_cast(float, x.x);
    return x.y == epoch;
}
__device__ __forceinline__ bool slot_load(__amdgpu_buffer_rsrc_t region, uint32_t off, uint32_t epoch, double &v)
{
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(region, (int)off, 0, kAuxSc1);
    v = __builtin_bit_cast(double, ((u64)x.z << 32) | x.x);
    return x.y == epoch && x.w == epoch;
}

// ... and written with ONE sc1 (write-through) buffer store: 288 eight-byte stores from one wave were the longest leg
// of a hand-off; a 16-byte sc1 store keeps each 8-byte half whole (MI355X_MICROARCH.md, Valid forms, R2).
__device__ __forceinline__ void slot_store(__amdgpu_buffer_rsrc_t region, uint32_t off, uint32_t epoch, float v)
{
    const u32x2 x = {__builtin_bit_cast(uint32_t, v), epoch};
    __builtin_amdgcn_raw_buffer_store_b64(x, region, (int)off, 0, kAuxSc1);
}
__device__ __forceinline__ void slot_store(__amdgpu_buffer_rsrc_t region, uint32_t off, uint32_t epoch, double v)
{
    const u64 b = __builtin_bit_cast(u64, v);
    const u32x4 x = {(uint32_t)b, epoch, (uint32_t)(b >> 32), epoch};
    __builtin_amdgcn_raw_buffer_store_b128(x, region, (int)off, 0, kAuxSc1);
}

// Bytes between the partial-product slots of two workgroups.  A slot of its own 128-byte line keeps 8 producers from
// writing into one line that every consumer polls.
#ifndef GBDPCG_PERSIST_PSTRIDE
#define GBDPCG_PERSIST_PSTRIDE 128
#endif
constexpr uint32_t kPartStrideWords = GBDPCG_PERSIST_PSTRIDE / 8;
#ifndef GBDPCG_PERSIST_SLEEP0
#define GBDPCG_PERSIST_SLEEP0 0
#endif

// Workspace of one problem, in u64 words: [ctrl | part[2][N] slots | halo[2][N][2][n] values | halo2 (same)], sized for one knot
// per workgroup (the largest workgroup count), whatever K a launch uses.
template <typename T> __host__ __device__ inline size_t persist_part_words(uint32_t N)
{
    return (size_t)2 * N * (kPartStrideWords > Gran<T>::PER ? kPartStrideWords : Gran<T>::PER);
}
template <typename T> __host__ __device__ inline size_t persist_halo_words(uint32_t n, uint32_t N)
{
    return (size_t)2 * N * 2 * n * Gran<T>::PER;
}
// ... followed by a second halo region of the same size, used by the single-reduction kernel only (its extra hand-off)
template <typename T> __host__ __device__ inline size_t persist_words(uint32_t n, uint32_t N)
{
    return kPersistCtrl + persist_part_words<T>(N) + 2 * persist_halo_words<T>(n, N);
}

// Wave 0 of a workgroup: poll until every partial of this epoch and the two neighbour boundary vectors have
// arrived (sc1 loads, no fence: the data is the flag), then sum the partials in a fixed order.  Every lane re-reads
// only the slots it is still missing and sleeps between passes: polling traffic competes with the very stores it is
// waiting for (measured on config 4: keeping two / three polls in flight per slot made an iteration 19 % / 36 % slower).
// part_off: byte offset of this epoch parity's partial slots in the region; hl_off / hr_off: byte offsets of the left /
// right neighbour's boundary vector (negative: no neighbour).  Returns false when the spin bound is hit.
#ifndef GBDPCG_PERSIST_GAP
#define GBDPCG_PERSIST_GAP 1
#endif
template <typename T, int NCT, uint32_t PJ, uint32_t NV, uint32_t HN = NCT>
__device__ __forceinline__ bool persist_sweep_w(__amdgpu_buffer_rsrc_t region, uint32_t part_off, int hl_off, int hr_off,
                                                uint32_t W, uint32_t epoch, uint32_t lane, uint32_t spin_limit, T (&total)[NV ? NV : 1],
                                                T *yl, T *yr)
{
    // NV values per partial slot, adjacent (0: neighbours only, nothing is summed); HN values per neighbour boundary
    constexpr uint32_t PER = Gran<T>::PER, PSTRIDE = (kPartStrideWords > PER ? kPartStrideWords : PER) * 8;
    constexpr uint32_t NS = NV ? PJ * NV : 1, HV = (HN + 63) / 64;
    static_assert(NV * PER * 8 <= PSTRIDE, "the values of a slot share its line");
    T pv[NS], hl[HV], hr[HV];
    bool have[NS], have_l[HV], have_r[HV];
#pragma unroll
    for (uint32_t j = 0; j < NS; ++j) {
        pv[j] = T(0);
        have[j] = NV == 0 || lane + 64 * (j / (NV ? NV : 1)) >= W;
    }
#pragma unroll
    for (uint32_t v = 0; v < HV; ++v) {
        hl[v] = hr[v] = T(0);
        have_l[v] = hl_off < 0 || lane + 64 * v >= HN;
        have_r[v] = hr_off < 0 || lane + 64 * v >= HN;
    }
    if (GBDPCG_PERSIST_SLEEP0) __builtin_amdgcn_s_sleep(GBDPCG_PERSIST_SLEEP0);
    // Two stages: the partials first (everybody's: the wait proper), the neighbours' boundary knots afterwards (published
    // at the same time, so normally one pass): the fewer loads are in flight during the wait, the sooner it ends.
    uint32_t spins = 0;
#ifdef GBDPCG_PERSIST_TWO_STAGE_ALWAYS
    constexpr bool TWO_STAGE = NV > 0;
#else
    constexpr bool TWO_STAGE = NV > 0 && NS + 2 * HV > 4;
#endif   // few loads per lane: one stage is as fast (measured, config 4)
    if constexpr (TWO_STAGE) {
        for (;; ++spins) {
            bool all = true;
#pragma unroll
            for (uint32_t j = 0; j < NS; ++j) {
                if (!have[j])
                    have[j] = slot_load(region, part_off + (lane + 64 * (j / (NV ? NV : 1))) * PSTRIDE + (j % (NV ? NV : 1)) * PER * 8,
                                        epoch, pv[j]);
                all = all && have[j];
            }
            if (__all(all)) break;
            if (spins >= spin_limit) return false;
            __builtin_amdgcn_s_sleep(GBDPCG_PERSIST_GAP);
        }
    }
    for (;; ++spins) {
        bool all = true;
        if constexpr (NV > 0 && !TWO_STAGE) {
#pragma unroll
            for (uint32_t j = 0; j < NS; ++j) {
                if (!have[j])
                    have[j] = slot_load(region, part_off + (lane + 64 * (j / (NV ? NV : 1))) * PSTRIDE + (j % (NV ? NV : 1)) * PER * 8,
                                        epoch, pv[j]);
                all = all && have[j];
            }
        }
#pragma unroll
        for (uint32_t v = 0; v < HV; ++v) {
            if (!have_l[v]) have_l[v] = slot_load(region, (uint32_t)hl_off + (lane + 64 * v) * PER * 8, epoch, hl[v]);
            if (!have_r[v]) have_r[v] = slot_load(region, (uint32_t)hr_off + (lane + 64 * v) * PER * 8, epoch, hr[v]);
            all = all && have_l[v] && have_r[v];
        }
        if (__all(all)) break;
        if (spins >= spin_limit) return false;
        __builtin_amdgcn_s_sleep(GBDPCG_PERSIST_GAP);
    }
    if constexpr (NV > 0) {
#pragma unroll
        for (uint32_t v = 0; v < NV; ++v) {
            T sum = pv[v];
#pragma unroll
            for (uint32_t j = 1; j < PJ; ++j) sum += pv[j * NV + v];
            total[v] = wave_sum(sum);
        }
    }
#pragma unroll
    for (uint32_t v = 0; v < HV; ++v)
        if (lane + 64 * v < HN) {
            yl[lane + 64 * v] = hl[v];
            yr[lane + 64 * v] = hr[v];
        }
    return true;
}
// partial slots per lane by workgroup count: the sums of the three forms differ only in how many zeros they add
template <typename T, int NCT, uint32_t NV, uint32_t HN = NCT>
__device__ __forceinline__ bool persist_sweep_n(__amdgpu_buffer_rsrc_t region, uint32_t part_off, int hl_off, int hr_off,
                                                uint32_t W, uint32_t epoch, uint32_t lane, uint32_t spin_limit,
                                                T (&total)[NV ? NV : 1], T *yl, T *yr)
{
    if (NV == 0 || W <= 64) return persist_sweep_w<T, NCT, 1, NV, HN>(region, part_off, hl_off, hr_off, W, epoch, lane, spin_limit, total, yl, yr);
    if (W <= 128) return persist_sweep_w<T, NCT, 2, NV, HN>(region, part_off, hl_off, hr_off, W, epoch, lane, spin_limit, total, yl, yr);
    return persist_sweep_w<T, NCT, 4, NV, HN>(region, part_off, hl_off, hr_off, W, epoch, lane, spin_limit, total, yl, yr);
}
template <typename T, int NCT>
__device__ __forceinline__ bool persist_sweep(__amdgpu_buffer_rsrc_t region, uint32_t part_off, int hl_off, int hr_off,
                                              uint32_t W, uint32_t epoch, uint32_t lane, uint32_t spin_limit, T &total,
                                              T *yl, T *yr)
{
    T tot[1];
    const bool ok = persist_sweep_n<T, NCT, 1>(region, part_off, hl_off, hr_off, W, epoch, lane, spin_limit, tot, yl, yr);
    total = tot[0];
    return ok;
}

// acc = sum_i m[i] * (A[i] + coef * B[i]) over this thread's COLS columns, ascending.  The operands come out of LDS
// in 16-byte pieces, STEP columns per step.  FENCED (three knots per workgroup: 960 threads, 128 VGPRs): the steps are
// pinned in order, because left alone hipcc hoists all 2 * COLS reads above the first multiply and spills; with fewer
// threads the hoisting is what we want (one LDS latency per product instead of one per step).
template <typename T, uint32_t COLS, bool FENCED, bool ALIGNED = true>
__device__ __forceinline__ T persist_row_dot(const T (&m)[COLS], const T *A, const T *B, T coef)
{
    T acc = T(0);
    if constexpr (ALIGNED) {
        constexpr uint32_t VW = 16 / sizeof(T), STEP = COLS % 4 == 0 ? 4 : VW;
        static_assert(COLS % STEP == 0 && STEP % VW == 0, "columns per thread come in whole 16-byte pieces");
        typedef T vec_t __attribute__((ext_vector_type(VW)));
#pragma unroll
        for (uint32_t i0 = 0; i0 < COLS; i0 += STEP) {
            T av[STEP], bv[STEP];
#pragma unroll
            for (uint32_t q = 0; q < STEP / VW; ++q) {
                const vec_t va = *reinterpret_cast<const vec_t *>(A + i0 + q * VW), vb = *reinterpret_cast<const vec_t *>(B + i0 + q * VW);
#pragma unroll
                for (uint32_t e = 0; e < VW; ++e) {
                    av[q * VW + e] = va[e];
                    bv[q * VW + e] = vb[e];
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < STEP; ++j) acc = fma_t(m[i0 + j], fma_t(coef, bv[j], av[j]), acc);
            if (FENCED) __builtin_amdgcn_sched_barrier(0);
        }
    } else {   // block sizes whose rows are not multiples of 16 bytes (n = 14 in fp32): element-wise LDS reads
#pragma unroll
        for (uint32_t i = 0; i < COLS; ++i) acc = fma_t(m[i], fma_t(coef, B[i], A[i]), acc);
    }
    return acc;
}

// Sum over the 8 lanes of an aligned group, in the VALU (DPP quad permutes + half mirror): every lane gets the total.
template <typename T> __device__ __forceinline__ T group_sum8(T v)
{
    v = dpp_add<0xB1>(v);    // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm:[2,3,0,1]
    return dpp_add<0x141>(v);  // row_half_mirror
}

enum PersistPhase { PP_INIT = 0, PP_PRECOND = 1, PP_DIRECTION = 2 };

// Diagnostic build only (-DGBDPCG_PERSIST_STAMPS): workgroup 1 leaves cycle stamps of the phases of iteration 3 in the
// unused control words of the workspace (read back by tools/persist_stamps.py).  No stamp exists in the shipped build.
#ifdef GBDPCG_PERSIST_STAMPS
#define GBDPCG_STAMP(IDX, COND)                                                                   \
    if ((COND) && lane == 0) ws[IDX] = __builtin_amdgcn_s_memtime();
#define GBDPCG_STAMP_RT(IDX, COND)                                                                \
    if ((COND) && lane == 0) ws[IDX] = __builtin_amdgcn_s_memrealtime();
// every workgroup's 100 MHz real-time clock at its publish (0) and at the end of its sweep (1), into the second halo region
#define GBDPCG_XSTAMP(WHICH, COND)                                                                \
    if ((COND) && lane == 0) (part + persist_part_words<T>(N) + persist_halo_words<T>(n, N))[2 * w + (WHICH)] = __builtin_amdgcn_s_memrealtime();
#else
#define GBDPCG_STAMP(IDX, COND)
#define GBDPCG_STAMP_RT(IDX, COND)
#define GBDPCG_XSTAMP(WHICH, COND)
#endif

// Lane map of one knot: aligned groups of 8 lanes share a row, lane g of the group holds columns [g*COLS, (g+1)*COLS) of
// that row of S and of Pinv in registers; a wave covers 8 rows, ceil(n/8) waves a knot (n = 36: 5 waves, 14 columns per
// lane, 56 VGPRs of fp64 matrix data).  The 8 partial sums of a row are folded with three DPP adds, so a product needs
// no LDS round trip and no barrier of its own.
template <typename T, int NCT> struct PersistGeom {
    static constexpr uint32_t n = NCT, G = 8, WPK = (n + 7) / 8, TPK = WPK * 64, VW = 16 / sizeof(T);
    static constexpr uint32_t CPL = (3 * n + G - 1) / G;   // columns per lane
    // 16-byte operand reads need the knots of a window and the lanes' column runs on 16-byte boundaries
    static constexpr bool ALIGNED = (n * sizeof(T)) % 16 == 0 && ((CPL + VW - 1) / VW * VW * sizeof(T)) % 16 == 0;
    static constexpr uint32_t COLS = ALIGNED ? (CPL + VW - 1) / VW * VW : CPL;
};

template <typename T, int NCT, int K, bool HAS_PINV>
__global__ __launch_bounds__((K * PersistGeom<T, NCT>::TPK)) void pcg_persist_kernel(PcgArgs<T> a, u64 *ws_all, uint32_t W,
                                                                                     uint32_t spin_limit, uint32_t staged,
                                                                                     uint32_t hold_wg, uint32_t hold_us)
{
    GBDPCG_PERSIST_HOLD(blockIdx.x % W, hold_wg, hold_us)
    (void)hold_wg;
    (void)hold_us;
    using Gm = PersistGeom<T, NCT>;
    constexpr uint32_t n = NCT, G = Gm::G, WPK = Gm::WPK, COLS = Gm::COLS, PER = Gran<T>::PER;
    // PUB publishes the partial, HPUB the boundary knots (no global store by the polling wave 0 when there are others)
    constexpr uint32_t THREADS = K * Gm::TPK, NWAVES = K * WPK, PUB = NWAVES > 1 ? 1 : 0, HPUB = NWAVES > 2 ? 2 : PUB;
    constexpr uint32_t WIN = (K + 2) * n, WINP = align16<T>(WIN + G * COLS - 3 * n + 1), OWN = K * n;
    static_assert(3 * n <= G * COLS && THREADS <= 1024, "lane map");

    __shared__ __attribute__((aligned(16))) T rwin[2][WINP];
    __shared__ __attribute__((aligned(16))) T pwin[2][WINP];
    __shared__ __attribute__((aligned(16))) T uwin[WINP];   // upsilon = S p, window form
    __shared__ __attribute__((aligned(16))) T twin[WINP];   // r~ = Pinv r, window form
    __shared__ __attribute__((aligned(16))) T lwin[WINP];   // lambda window of the prologue
    __shared__ __attribute__((aligned(16))) T lam[OWN];
    __shared__ T dots[NWAVES];   // per-wave shares of the inner product
    __shared__ T bc[4];          // [0] coefficient of the next phase, [1] eta, [2] beta of the last direction update
    __shared__ uint32_t bci[4];  // [0] stop (1 converged, 2 hand-off timed out), [1] iterations, [3] persist_leave's flag
    __shared__ T rescue_red[2 * NWAVES];

    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t slot = wave / WPK, wv = wave - slot * WPK;
    const uint32_t g = lane & 7u, row = wv * 8 + (lane >> 3);
    const bool row_live = row < n;
    const uint32_t cbase = g * COLS;
    const uint32_t oi = slot * n + (row_live ? row : 0u);   // this lane's row in the own part of a vector

    // chunk index: blocks b and b + 8 share an XCD (round-robin dispatch), so consecutive chunks -- the ones that
    // exchange halo knots -- are put on one XCD.  Speed only: the protocol does not depend on placement.
    const uint32_t prob = blockIdx.x / W, b = blockIdx.x - prob * W;
    const uint32_t w = (W % 8 == 0) ? (b % 8) * (W / 8) + b / 8 : b;
    const uint32_t k0 = w * K, k = k0 + slot;
    const bool knot_live = k < N;

    const size_t mstride = (size_t)3 * n * n * N;
    const T *S = a.S + prob * mstride;
    const T *P = HAS_PINV ? a.Pinv + prob * mstride : nullptr;
    const T *gamma = a.gamma + (size_t)prob * len;
    T *lambda = a.lambda + (size_t)prob * len;

    u64 *ws = ws_all + (size_t)prob * persist_words<T>(n, N);
    constexpr uint32_t PSW = kPartStrideWords > PER ? kPartStrideWords : PER;   // u64 words per partial slot
    u64 *part = ws + kPersistCtrl;                              // [2][N] slots
    // the same words as a buffer resource for the polling loads (byte offsets from `part`)
    const __amdgpu_buffer_rsrc_t region = __builtin_amdgcn_make_buffer_rsrc(
        part, 0, (int)((persist_words<T>(n, N) - kPersistCtrl) * 8), 0x00020000);
    const uint32_t halo_base = (uint32_t)(persist_part_words<T>(N) * 8);
    const uint32_t base = (uint32_t)__hip_atomic_load(ws, GBDPCG_RLX_AGENT);   // epochs of this launch continue from here
    GBDPCG_PERSIST_DROP(ws, W, base, 2u * a.max_iter + 8u, hold_us)

    // ---- resident matrices: this lane's COLS columns of its row, both matrices -----------------------------------
    T sreg[COLS], preg[COLS];
    {
        const T *Sk = S + (size_t)(knot_live ? k : 0u) * 3 * n * n;
        const T *Pk = (HAS_PINV ? P : S) + (size_t)(knot_live ? k : 0u) * 3 * n * n;
        // all 2 * COLS loads first (one memory round trip), the selections afterwards
        T sraw[COLS], praw[COLS];
        if (staged) {
            // The block-rows of the workgroup's K knots are contiguous in memory: LDS-DMA brings them in as dense 16-byte
            // pieces (the direct form below reads 8 bytes per lane at a stride of n elements) and the lanes pick their
            // elements up from LDS.  Knots past the end of the problem re-read the last one (masked below).
            extern __shared__ __attribute__((aligned(16))) unsigned char persist_stage[];
            constexpr uint32_t PPK = 3 * n * n * sizeof(T) / 16;   // 16-byte pieces per knot
            static_assert((3 * n * n * sizeof(T)) % 16 == 0, "whole pieces per knot");
            T *stS = reinterpret_cast<T *>(persist_stage), *stP = stS + K * 3 * n * n;
            uint32_t lo = lane;
            asm volatile("" : "+v"(lo));
            for (uint32_t q0 = wave * 64; q0 < K * PPK; q0 += THREADS) {
                const uint32_t q = q0 + lo;
                if (q < K * PPK) {
                    const uint32_t j = q / PPK, piece = q - j * PPK, kk = k0 + j < N ? k0 + j : N - 1;
                    const uint32_t off = (kk - (k0 < N ? k0 : N - 1)) * (3 * n * n * (uint32_t)sizeof(T)) + piece * 16;
                    const T *Sb = S + (size_t)(k0 < N ? k0 : N - 1) * 3 * n * n;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(Sb), "v"(off), "s"((uint32_t)(uintptr_t)stS + q0 * 16) : "memory");
                    if (HAS_PINV) {
                        const T *Pb = P + (size_t)(k0 < N ? k0 : N - 1) * 3 * n * n;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "s"(Pb), "v"(off), "s"((uint32_t)(uintptr_t)stP + q0 * 16) : "memory");
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (uint32_t i = 0; i < COLS; ++i) {
                const uint32_t c = cbase + i;
                const bool valid = row_live && knot_live && c < 3 * n && !(k == 0 && c < n) && !(k == N - 1 && c >= 2 * n);
                const uint32_t idx = slot * 3 * n * n + (valid ? c * n + row : n * n);
                sraw[i] = stS[idx];
                if (HAS_PINV) praw[i] = stP[idx];
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < COLS; ++i) {
                const uint32_t c = cbase + i;
                const bool valid = row_live && knot_live && c < 3 * n && !(k == 0 && c < n) && !(k == N - 1 && c >= 2 * n);
                const uint32_t idx = valid ? c * n + row : n * n;   // else an element of D_k: always there
                sraw[i] = Sk[idx];
                if (HAS_PINV) praw[i] = Pk[idx];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t i = 0; i < COLS; ++i) {
            const uint32_t c = cbase + i;
            // L_0 and R_{N-1} are never used (pcg.cuh:105-106): zero, whatever the storage holds
            const bool valid = row_live && knot_live && c < 3 * n && !(k == 0 && c < n) && !(k == N - 1 && c >= 2 * n);
            sreg[i] = valid ? sraw[i] : T(0);
            if (HAS_PINV) preg[i] = valid ? praw[i] : T(0);
            else preg[i] = (valid && c == n + row) ? T(1) : T(0);   // d_Pinv == NULL: identity preconditioner
        }
    }

    // ---- LDS state ---------------------------------------------------------------------------------------------
    for (uint32_t i = tid; i < WINP; i += THREADS) {
        rwin[0][i] = rwin[1][i] = pwin[0][i] = pwin[1][i] = uwin[i] = twin[i] = T(0);
        const int64_t gi = (int64_t)k0 * n - n + i;   // window element i = vector element gi
        lwin[i] = (i < WIN && gi >= 0 && gi < (int64_t)len) ? lambda[gi] : T(0);
    }
    for (uint32_t i = tid; i < OWN; i += THREADS) lam[i] = (k0 * n + i < len) ? lambda[k0 * n + i] : T(0);
    if (tid == 0) {
        bc[0] = bc[1] = bc[2] = T(0);
        bci[0] = bci[1] = 0u;
    }
    __syncthreads();

    uint32_t rc = 0, pc = 0;   // current r / p window

    // One phase: y = M X over the own knots with X = A + coef * B formed on the fly; the row owners (lane 0 of each
    // group) store y, the new operand window NEW, publish the boundary knots and their share of the inner product; then
    // wave 0 publishes the workgroup's partial, gathers everybody's and decides what comes next.
#define GBDPCG_PERSIST_PHASE(PHASE, MREG, A, B, NEW, YWIN, EPOCH, ITER)                                              \
    {                                                                                                                \
        const T coef = bc[0];                                                                                        \
        const uint32_t tag = base + (EPOCH), par = (EPOCH) & 1u;                                                     \
        const bool stamp_here = w == 1 && wave == 0 && (ITER) == 3;                                                  \
        const uint32_t sb = PHASE == PP_DIRECTION ? 2u : 8u;                                                         \
        (void)stamp_here;                                                                                            \
        (void)sb;                                                                                                    \
        GBDPCG_STAMP(sb + 0, stamp_here)                                                                             \
        GBDPCG_STAMP_RT(PHASE == PP_DIRECTION ? 14u : 15u, stamp_here)                                               \
        /* operands of the row owners and the two halo knots of the new window first: their LDS round trips run   \
           under the product's */                                                                                    \
        const T own_a = (A)[n + oi], own_b = (B)[n + oi], own_p = pwin[pc][n + oi], own_l = lam[oi];                  \
        if ((NEW) != nullptr) {                                                                                      \
            for (uint32_t i = tid; i < 2 * n; i += THREADS) {                                                        \
                const uint32_t j = i < n ? i : OWN + i;                                                              \
                (NEW)[j] = fma_t(coef, (B)[j], (A)[j]);                                                              \
            }                                                                                                        \
        }                                                                                                            \
        T y = persist_row_dot<T, COLS, (THREADS > 768), Gm::ALIGNED>(MREG, (A) + slot * n + cbase, (B) + slot * n + cbase, coef);   \
        GBDPCG_STAMP(1, stamp_here && PHASE == PP_DIRECTION)                                                         \
        y = group_sum8(y);                                                                                           \
        T d = T(0);                                                                                                  \
        if (g == 0 && row_live) {                                                                                    \
            const T xo = fma_t(coef, own_b, own_a);   /* this row of the operand */                                  \
            if (PHASE == PP_INIT) y = (knot_live ? gamma[k * n + row] : T(0)) - y;   /* r = gamma - S lambda */      \
            (YWIN)[n + oi] = y;                                                                                      \
            if ((NEW) != nullptr) (NEW)[n + oi] = xo;                                                                \
            if (PHASE == PP_PRECOND && (NEW) != nullptr)   /* lambda += alpha p   (pcg.cuh:172-174); coef = -alpha */ \
                lam[oi] = fma_t(-coef, own_p, own_l);                                                                \
            if (PHASE != PP_INIT) d = xo * y;                                                                        \
        }                                                                                                            \
        d = wave_sum(d);                                                                                             \
        if (lane == 0) dots[wave] = d;                                                                               \
        GBDPCG_STAMP(7, stamp_here && PHASE == PP_DIRECTION)                                                         \
        GBDPCG_STAMP(13, stamp_here && PHASE == PP_DIRECTION)                                                        \
        __syncthreads();                                                                                             \
        GBDPCG_STAMP(sb + 1, stamp_here)                                                                             \
        /* No global store is issued before the barrier above (it would make every wave wait for the write-through  \
           acknowledgement), and none by the polling wave (its loads would queue behind them): wave PUB publishes. */  \
        if (wave == PUB) {   /* the partial first: every workgroup waits for it, the boundary knots concern two */   \
            T dot = dots[0];                                                                                         \
            _Pragma("unroll") for (uint32_t q = 1; q < NWAVES; ++q) dot += dots[q];                                 \
            GBDPCG_XSTAMP(0, PHASE == PP_DIRECTION && (ITER) == 3)                                                   \
            if (lane == 0) slot_store(region, (par * N + w) * PSW * 8u, tag, dot);                                   \
        }                                                                                                            \
        if (wave == HPUB) {                                                                                          \
            const uint32_t my_halo = halo_base + ((par * N + w) * 2) * n * PER * 8u;                                 \
            for (uint32_t i = lane; i < 2 * n; i += 64) {   /* first knot -> left neighbour, last knot -> right */     \
                const uint32_t src = i < n ? n + i : n + (K - 1) * n + (i - n);                                       \
                slot_store(region, my_halo + i * PER * 8u, tag, (YWIN)[src]);                                        \
            }                                                                                                        \
        }                                                                                                            \
        if (wave == 0) {                                                                                             \
            GBDPCG_STAMP(sb + 2, stamp_here)                                                                         \
            T total;                                                                                                 \
            const bool ok = persist_sweep<T, NCT>(                                                                   \
                region, par * N * PSW * 8u,                                                                          \
                w > 0 ? (int)(halo_base + ((par * N + (w - 1)) * 2 + 1) * n * PER * 8u) : -1,                        \
                w + 1 < W ? (int)(halo_base + ((par * N + (w + 1)) * 2) * n * PER * 8u) : -1, W, tag, lane,          \
                spin_limit, total, (YWIN), (YWIN) + n + OWN);                                                        \
            GBDPCG_STAMP(sb + 3, stamp_here)                                                                         \
            GBDPCG_XSTAMP(1, PHASE == PP_DIRECTION && (ITER) == 3)                                                   \
            if (lane == 0) {                                                                                         \
                if (!ok) {                                                                                           \
                    bci[0] = 2u;                                                                                     \
                } else if (PHASE == PP_INIT) {                                                                       \
                    bc[0] = T(0);                                                                                    \
                } else if (PHASE == PP_DIRECTION) {        /* alpha = eta / (p . upsilon)      (pcg.cuh:169) */      \
                    bc[0] = -(bc[1] / total);                                                                        \
                } else if ((ITER) < 0) {                    /* prologue: eta = r . r~, p = r~  (pcg.cuh:139-149) */   \
                    bc[1] = total;                                                                                   \
                    bc[0] = bc[2] = T(0);                                                                            \
                } else if (fabs(total) < a.tol) {           /* pcg.cuh:195 */                                         \
                    bci[0] = 1u;                                                                                     \
                    bci[1] = (uint32_t)(ITER) + 1u;                                                                  \
                } else {                                    /* beta = eta' / eta ; eta = eta'  (pcg.cuh:199-202) */   \
                    bc[0] = bc[2] = total / bc[1];                                                                   \
                    bc[1] = total;                                                                                   \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        __syncthreads();                                                                                             \
        GBDPCG_STAMP(sb + 4, stamp_here)                                                                             \
    }

    T *const none = nullptr;
    // r = gamma - S lambda (pcg.cuh:118-126); the boundary knots of r travel with epoch 1
    GBDPCG_PERSIST_PHASE(PP_INIT, sreg, lwin, lwin, none, rwin[0], 1u, -1)
    // r~ = Pinv r ; eta = r . r~ (pcg.cuh:130-149)
    if (bci[0] == 0u) GBDPCG_PERSIST_PHASE(PP_PRECOND, preg, rwin[0], rwin[0], none, twin, 2u, -1)

    uint32_t iter = 0;
    bool ran_out = true;
    for (; iter < a.max_iter && bci[0] == 0u; ++iter) {   // pcg.cuh:154
        // p = r~ + beta p ; upsilon = S p ; alpha = eta / (p . upsilon)   (pcg.cuh:203-206,156-169)
        GBDPCG_PERSIST_PHASE(PP_DIRECTION, sreg, twin, pwin[pc], pwin[pc ^ 1], uwin, 3u + 2u * iter, (int)iter)
        pc ^= 1u;
        if (bci[0] != 0u) break;
        // lambda += alpha p ; r -= alpha upsilon ; r~ = Pinv r ; eta' = r . r~   (pcg.cuh:172-193)
        GBDPCG_PERSIST_PHASE(PP_PRECOND, preg, rwin[rc], uwin, rwin[rc ^ 1], twin, 4u + 2u * iter, (int)iter)
        rc ^= 1u;
        if (bci[0] == 1u) {
            ran_out = false;
            break;
        }
    }
#undef GBDPCG_PERSIST_PHASE

    // ---- outputs (pcg.cuh:212,215) --------------------------------------------------------------------------------
    const bool failed = bci[0] == 2u;   // (the same verdict in every workgroup of the problem: all of them wait for all)
    const T beta = ran_out ? bc[2] : T(0);   // the last iteration did not break: it still ran p = r~ + beta p
    for (uint32_t i = tid; i < OWN; i += THREADS) {
        const uint32_t gi = k0 * n + i;
        if (gi < len && !failed) {   // a launch that gave up leaves lambda, r, p as it found them: the rescue launch starts there
            lambda[gi] = lam[i];
            if (a.r) a.r[(size_t)prob * len + gi] = rwin[rc][n + i];
            if (a.p) a.p[(size_t)prob * len + gi] = ran_out ? fma_t(beta, pwin[pc][n + i], twin[n + i]) : pwin[pc][n + i];
        }
    }
    if (w == 0 && tid == 0 && (!failed || a.rescue_off)) {   // (rescue_off: hooks build only, to show the mark)
        a.iters[prob] = failed ? kItersGaveUp : (ran_out ? a.max_iter : bci[1]);
        if (a.max_iter_exit) a.max_iter_exit[prob] = failed ? 2 : (ran_out ? 1 : 0);
    }
    persist_leave<T, NWAVES>(a, prob, ws, W, base, 2u * a.max_iter + 8u, failed, rescue_red, bci + 3);
}

// ---- single-reduction variant (opt-in: GBDPCG_PATH_PERSISTENT_1R) ------------------------------------------------------
// The same solve in the Chronopoulos-Gear form of preconditioned CG, arranged so that an iteration crosses the chip
// ONCE (the north star's "single cross-CU reduction per iteration"):
//     u = Pinv r ; w = S u ; gamma = r.u ; delta = u.w                      (both inner products in ONE all-gather)
//     beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old)      (beta = 0, alpha = gamma / delta first)
//     p = u + beta p ; s = w + beta s ; lambda += alpha p ; r -= alpha s
// w = S u needs u on the two knots next to the workgroup's own.  Instead of a second hand-off they are recomputed here:
// the workgroup also keeps the Pinv block-rows of those two knots in registers (the threads of its first / last own
// knot hold one more row run each: +28 VGPRs in fp64) and carries r, s and w on a TWO-knot halo; what travels with the
// all-gather is {gamma, delta} and the two outer own knots of w on each side.  Redundant values are bit-identical: every
// copy is the same sequence of fma's on the same bits.
// In exact arithmetic p, lambda, r and the tested quantity gamma_i = r_i . Pinv r_i are those of pcg.cuh:154-206 (the exit
// test |gamma| < tol is the same test, seen one product pair later); the ROUNDING sequence differs (alpha is not
// eta / (p . S p), and s = S p is carried by recurrence), so this is not the reference's recurrence and AUTO does not
// pick it.  Against the oracle: equal iteration counts and fp64 lambda within 1e-10 on every shape of
// tests/test_gpu_persist.py (a = 0.5 and a = 0.9 generators).  Two or three knots per workgroup only.
template <typename T, int NCT, int K, bool HAS_PINV>
__global__ __launch_bounds__((K * PersistGeom<T, NCT>::TPK)) void pcg_persist1r_kernel(PcgArgs<T> a, u64 *ws_all, uint32_t W,
                                                                                       uint32_t spin_limit, uint32_t hold_wg,
                                                                                       uint32_t hold_us)
{
    GBDPCG_PERSIST_HOLD(blockIdx.x % W, hold_wg, hold_us)
    (void)hold_wg;
    (void)hold_us;
    using Gm = PersistGeom<T, NCT>;
    static_assert(K >= 2, "the first and the last own knot carry one halo block-row each");
    constexpr uint32_t n = NCT, G = Gm::G, WPK = Gm::WPK, COLS = Gm::COLS, PER = Gran<T>::PER;
    constexpr uint32_t THREADS = K * Gm::TPK, NWAVES = K * WPK, PUB = 1, HPUB = 2, HPUB2 = 3;   // publishing waves
    static_assert(NWAVES >= 4, "wave 0 polls, three others publish");
    // windows cover knots k0-2 .. k0+K+1: window knot j = vector knot k0 - 2 + j, own knots are j = 2 .. K+1
    constexpr uint32_t WIN = (K + 4) * n, WINP = align16<T>(WIN + G * COLS - 3 * n + 1), OWN = K * n, HN = 2 * n;
    static_assert(3 * n <= G * COLS && THREADS <= 1024, "lane map");

    __shared__ __attribute__((aligned(16))) T rwin[WINP], swin[WINP], uwin[WINP], wwin[WINP], lwin[WINP];
    __shared__ __attribute__((aligned(16))) T lam[OWN], pown[OWN];
    __shared__ T dots_g[NWAVES], dots_d[NWAVES];
    __shared__ T bc[4];          // [0] alpha, [1] beta, [2] gamma_old, [3] alpha_old
    __shared__ uint32_t bci[4];  // [0] stop (1 converged, 2 hand-off timed out, 3 ran out), [1] iterations, [3] persist_leave's flag
    __shared__ T rescue_red[2 * NWAVES];

    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t slot = wave / WPK, wv = wave - slot * WPK;
    const uint32_t g = lane & 7u, row = wv * 8 + (lane >> 3);
    const bool row_live = row < n;
    const uint32_t cbase = g * COLS;

    const uint32_t prob = blockIdx.x / W, b = blockIdx.x - prob * W;
    const uint32_t w = (W % 8 == 0) ? (b % 8) * (W / 8) + b / 8 : b;   // neighbours on one XCD (speed only)
    const uint32_t k0 = w * K, k = k0 + slot;
    // the halo knot whose Pinv block-row this thread carries as well: left of the first own knot / right of the last
    const bool has_halo = slot == 0 || slot == K - 1;
    const int64_t kh = slot == 0 ? (int64_t)k0 - 1 : (int64_t)k0 + K;
    const uint32_t jh = slot == 0 ? 1u : K + 2u;   // its window knot

    const size_t mstride = (size_t)3 * n * n * N;
    const T *S = a.S + prob * mstride;
    const T *P = HAS_PINV ? a.Pinv + prob * mstride : nullptr;
    const T *gamma = a.gamma + (size_t)prob * len;
    T *lambda = a.lambda + (size_t)prob * len;

    u64 *ws = ws_all + (size_t)prob * persist_words<T>(n, N);
    constexpr uint32_t PSW = kPartStrideWords > 2 * PER ? kPartStrideWords : 2 * PER;
    static_assert(PSW == kPartStrideWords, "two values per partial slot fit the slot stride");
    u64 *part = ws + kPersistCtrl;                       // [2][N] slots of {gamma, delta}
    const __amdgpu_buffer_rsrc_t region = __builtin_amdgcn_make_buffer_rsrc(
        part, 0, (int)((persist_words<T>(n, N) - kPersistCtrl) * 8), 0x00020000);
    const uint32_t h_base = (uint32_t)(persist_part_words<T>(N) * 8);
    const uint32_t base = (uint32_t)__hip_atomic_load(ws, GBDPCG_RLX_AGENT);
    GBDPCG_PERSIST_DROP(ws, W, base, 2u * a.max_iter + 8u, hold_us)

    // ---- resident block-rows: S and Pinv of the own knot, Pinv of the thread's halo knot ---------------------------
    T sreg[COLS], preg[COLS], hreg[COLS];
    {
        auto col_ok = [&](int64_t kk, uint32_t c) {   // L_0 and R_{N-1} are never used (pcg.cuh:105-106); knots outside [0, N) are zero
            return row_live && kk >= 0 && kk < (int64_t)N && c < 3 * n && !(kk == 0 && c < n) && !(kk == (int64_t)N - 1 && c >= 2 * n);
        };
        const uint32_t kc = k < N ? k : 0u, khc = (kh >= 0 && kh < (int64_t)N) ? (uint32_t)kh : 0u;
        const T *Sk = S + (size_t)kc * 3 * n * n;
        const T *Pk = (HAS_PINV ? P : S) + (size_t)kc * 3 * n * n;
        const T *Ph = (HAS_PINV ? P : S) + (size_t)khc * 3 * n * n;
        T sraw[COLS], praw[COLS], hraw[COLS];
#pragma unroll
        for (uint32_t i = 0; i < COLS; ++i) {
            const uint32_t c = cbase + i;
            sraw[i] = Sk[col_ok(k, c) ? c * n + row : n * n];
            if (HAS_PINV) {
                praw[i] = Pk[col_ok(k, c) ? c * n + row : n * n];
                hraw[i] = Ph[(has_halo && col_ok(kh, c)) ? c * n + row : n * n];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t i = 0; i < COLS; ++i) {
            const uint32_t c = cbase + i;
            const bool vk = col_ok(k, c), vh = has_halo && col_ok(kh, c);
            sreg[i] = vk ? sraw[i] : T(0);
            if (HAS_PINV) {
                preg[i] = vk ? praw[i] : T(0);
                hreg[i] = vh ? hraw[i] : T(0);
            } else {   // d_Pinv == NULL: identity preconditioner
                preg[i] = (vk && c == n + row) ? T(1) : T(0);
                hreg[i] = (vh && c == n + row) ? T(1) : T(0);
            }
        }
    }

    for (uint32_t i = tid; i < WINP; i += THREADS) {
        rwin[i] = swin[i] = uwin[i] = wwin[i] = T(0);
        const int64_t gi = (int64_t)k0 * n - 2 * (int64_t)n + i;
        lwin[i] = (i < WIN && gi >= 0 && gi < (int64_t)len) ? lambda[gi] : T(0);
    }
    for (uint32_t i = tid; i < OWN; i += THREADS) {
        lam[i] = (k0 * n + i < len) ? lambda[k0 * n + i] : T(0);
        pown[i] = T(0);
    }
    if (tid == 0) {
        bc[0] = bc[1] = bc[2] = bc[3] = T(0);
        bci[0] = bci[1] = 0u;
    }
    __syncthreads();

    // hand-off offsets (bytes from `part`): a workgroup publishes [side 0: its first two own knots | side 1: its last two]
    // hand-off offsets (bytes from `part`): a workgroup publishes [side 0: its first two own knots | side 1: its last two]
    auto nbr_l = [&](uint32_t par) { return w > 0 ? (int)(h_base + ((par * W + (w - 1)) * 2 + 1) * HN * PER * 8u) : -1; };
    auto nbr_r = [&](uint32_t par) { return w + 1 < W ? (int)(h_base + ((par * W + (w + 1)) * 2) * HN * PER * 8u) : -1; };
    // waves HPUB / HPUB2: the two outer own knots of YWIN on the left / right side, for the neighbours' two-knot halo
#define GBDPCG_1R_PUBLISH_HALO(YWIN, PAR, TAG)                                                                    \
    if (wave == HPUB || wave == HPUB2) {                                                                          \
        const uint32_t side = wave == HPUB ? 0u : 1u;                                                             \
        const uint32_t mh = h_base + (((PAR) * W + w) * 2 + side) * HN * PER * 8u;                                \
        for (uint32_t i = lane; i < HN; i += 64)                                                                  \
            slot_store(region, mh + i * PER * 8u, (TAG), (YWIN)[(side ? 2 * n + (K - 2) * n : 2 * n) + i]);       \
    }

    // r = gamma - S lambda (pcg.cuh:118-126) on the own knots; its two outer own knots go to the neighbours (epoch 1)
    {
        T y = group_sum8(persist_row_dot<T, COLS, (THREADS > 768), Gm::ALIGNED>(sreg, lwin + (slot + 1) * n + cbase,
                                                                               lwin + (slot + 1) * n + cbase, T(0)));
        if (g == 0 && row_live) rwin[(slot + 2) * n + row] = (k < N ? gamma[k * n + row] : T(0)) - y;
        __syncthreads();
        GBDPCG_1R_PUBLISH_HALO(rwin, 1u, base + 1u)
        if (wave == 0) {
            T dummy[1];
            const bool ok = persist_sweep_n<T, NCT, 0, HN>(region, 0u, nbr_l(1u), nbr_r(1u), W, base + 1u, lane, spin_limit, dummy,
                                                          rwin, rwin + (K + 2) * n);
            if (lane == 0 && !ok) bci[0] = 2u;
        }
        __syncthreads();
    }

    uint32_t iter = 0;
    for (; bci[0] == 0u; ++iter) {
        const uint32_t ew = 2u + iter, par = iter & 1u;
        const bool stamp_here = w == 1 && wave == 0 && iter == 3;
        (void)stamp_here;
        GBDPCG_STAMP(2, stamp_here)
        // u = Pinv r on the own knots and, redundantly, on the two knots next to them; share of gamma = r . u  (pcg.cuh:180-187)
        {
            T y = group_sum8(persist_row_dot<T, COLS, (THREADS > 768), Gm::ALIGNED>(preg, rwin + (slot + 1) * n + cbase,
                                                                                   rwin + (slot + 1) * n + cbase, T(0)));
            T yh = T(0);
            if (has_halo)   // wave-uniform
                yh = group_sum8(persist_row_dot<T, COLS, (THREADS > 768), Gm::ALIGNED>(hreg, rwin + (jh - 1) * n + cbase,
                                                                                       rwin + (jh - 1) * n + cbase, T(0)));
            T d = T(0);
            if (g == 0 && row_live) {
                uwin[(slot + 2) * n + row] = y;
                if (has_halo) uwin[jh * n + row] = yh;
                d = rwin[(slot + 2) * n + row] * y;
            }
            d = wave_sum(d);
            if (lane == 0) dots_g[wave] = d;
        }
        __syncthreads();
        GBDPCG_STAMP(3, stamp_here)
        // w = S u on the own knots ; share of delta = u . w
        {
            T y = group_sum8(persist_row_dot<T, COLS, (THREADS > 768), Gm::ALIGNED>(sreg, uwin + (slot + 1) * n + cbase,
                                                                                   uwin + (slot + 1) * n + cbase, T(0)));
            T d = T(0);
            if (g == 0 && row_live) {
                wwin[(slot + 2) * n + row] = y;
                d = uwin[(slot + 2) * n + row] * y;
            }
            d = wave_sum(d);
            if (lane == 0) dots_d[wave] = d;
        }
        __syncthreads();
        GBDPCG_STAMP(4, stamp_here)
        // {gamma, delta, the two outer own knots of w per side} in ONE all-gather
        if (wave == PUB) {   // the partials first: every workgroup waits for them
            T pg = dots_g[0], pd = dots_d[0];
#pragma unroll
            for (uint32_t q = 1; q < NWAVES; ++q) {
                pg += dots_g[q];
                pd += dots_d[q];
            }
            if (lane == 0) {
                slot_store(region, (par * N + w) * PSW * 8u, base + ew, pg);
                slot_store(region, (par * N + w) * PSW * 8u + PER * 8u, base + ew, pd);
            }
        }
        GBDPCG_1R_PUBLISH_HALO(wwin, par, base + ew)
        if (wave == 0) {
            T tot[2];
            const bool ok = persist_sweep_n<T, NCT, 2, HN>(region, par * N * PSW * 8u, nbr_l(par), nbr_r(par), W, base + ew, lane,
                                                          spin_limit, tot, wwin, wwin + (K + 2) * n);
            GBDPCG_STAMP(5, stamp_here)
            if (lane == 0) {
                const T gam = tot[0], del = tot[1];
                if (!ok) {
                    bci[0] = 2u;
                } else if (iter > 0 && fabs(gam) < a.tol) {       // pcg.cuh:195, on gamma_iter = r . Pinv r after `iter` updates
                    bci[0] = 1u;
                    bci[1] = iter;
                } else if (iter >= a.max_iter) {
                    bci[0] = 3u;
                    bci[1] = a.max_iter;
                    bc[1] = iter > 0 ? gam / bc[2] : T(0);        // the beta of the direction update the reference still runs
                } else {
                    const T beta = iter > 0 ? gam / bc[2] : T(0);
                    const T alpha = iter > 0 ? gam / (del - beta * gam / bc[3]) : gam / del;
                    bc[0] = alpha;
                    bc[1] = beta;
                    bc[2] = gam;
                    bc[3] = alpha;
                }
            }
        }
        __syncthreads();
        GBDPCG_STAMP(6, stamp_here)
        if (bci[0] != 0u) break;
        // s = w + beta s ; r -= alpha s on the own knots and the two-knot halo ; p = u + beta p ; lambda += alpha p on the own
        {
            const T alpha = bc[0], beta = bc[1];
            for (uint32_t i = tid; i < WIN; i += THREADS) {
                const T sn = fma_t(beta, swin[i], wwin[i]);
                swin[i] = sn;
                rwin[i] = fma_t(-alpha, sn, rwin[i]);
                if (i >= 2 * n && i < 2 * n + OWN) {
                    const T pn = fma_t(beta, pown[i - 2 * n], uwin[i]);
                    pown[i - 2 * n] = pn;
                    lam[i - 2 * n] = fma_t(alpha, pn, lam[i - 2 * n]);
                }
            }
        }
        __syncthreads();
        GBDPCG_STAMP(7, stamp_here)
    }
#undef GBDPCG_1R_PUBLISH_HALO

    // ---- outputs (pcg.cuh:212,215) --------------------------------------------------------------------------------
    const uint32_t stop = bci[0];
    const bool failed = stop == 2u, ran_out = stop == 3u;
    const T beta = ran_out ? bc[1] : T(0);   // max-iteration exit: the reference's last iteration still ran p = r~ + beta p
    for (uint32_t i = tid; i < OWN; i += THREADS) {
        const uint32_t gi = k0 * n + i;
        if (gi < len && !failed) {
            lambda[gi] = lam[i];
            if (a.r) a.r[(size_t)prob * len + gi] = rwin[2 * n + i];
            if (a.p) a.p[(size_t)prob * len + gi] = ran_out ? fma_t(beta, pown[i], uwin[2 * n + i]) : pown[i];
        }
    }
    if (w == 0 && tid == 0 && (!failed || a.rescue_off)) {
        a.iters[prob] = failed ? kItersGaveUp : bci[1];
        if (a.max_iter_exit) a.max_iter_exit[prob] = failed ? 2 : (ran_out ? 1 : 0);
    }
    persist_leave<T, NWAVES>(a, prob, ws, W, base, 2u * a.max_iter + 8u, failed, rescue_red, bci + 3);
}

// ---- launch of one instantiation (host) ----------------------------------------------------------------------------

template <typename T, int NCT, int K>
static hipError_t launch_persist_k(const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction)
{
    const uint32_t W = (a.N + K - 1) / K;
    // ~1 us per failed pass: a launch whose workgroups are not all resident gives up after about two seconds.
    uint32_t spin_limit = 1u << 21, hold_wg = 0u, hold_us = 0u;
#ifdef GBDPCG_TEST_HOOKS
    // variants/libgbdpcg_hooks.so only (tests/test_gpu_persist.py): a short spin bound, a launch that is missing its last
    // workgroup, a workgroup that arrives late -- to drive the give-up path.  The shipped library has none of this.
    if (const char *e = getenv("GBDPCG_PERSIST_SPIN_LIMIT")) spin_limit = (uint32_t)atoi(e);
    if (const char *e = getenv("GBDPCG_PERSIST_HOLD_US")) {
        hold_us = (uint32_t)atoi(e);
        hold_wg = W - 1u;
    }
    if (getenv("GBDPCG_PERSIST_DROP_WG")) hold_us = 0xffffffffu;
    const bool rescue_off = getenv("GBDPCG_RESCUE_OFF") != nullptr;   // show a test what a launch that gave up leaves behind
#else
    const bool rescue_off = false;
#endif
    PcgArgs<T> ka = a;
    ka.rescue_off = rescue_off;
    const dim3 grid(W * a.batch), block(K * PersistGeom<T, NCT>::TPK);
    u64 *ws = reinterpret_cast<u64 *>(workspace);
    if (one_reduction) {
        if constexpr (K >= 2) {
            if (a.Pinv) hipLaunchKernelGGL((pcg_persist1r_kernel<T, NCT, K, true>), grid, block, 0, s, ka, ws, W, spin_limit, hold_wg, hold_us);
            else hipLaunchKernelGGL((pcg_persist1r_kernel<T, NCT, K, false>), grid, block, 0, s, ka, ws, W, spin_limit, hold_wg, hold_us);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        // staged matrix loads (LDS-DMA) when both block-row sets of a workgroup fit the dynamic LDS next to the windows
        // and the matrices are 16-byte aligned
        static const bool no_staging = getenv("GBDPCG_PERSIST_DIRECT_LOADS") != nullptr;   // tuning runs only
        const size_t stage = (size_t)(a.Pinv ? 2 : 1) * K * 3 * NCT * NCT * sizeof(T);
        const bool staged = !no_staging && stage <= 128 * 1024 && !(reinterpret_cast<uintptr_t>(a.S) % 16) &&
                            !(a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % 16);
        const size_t lds = staged ? stage : 0;
        auto kern = a.Pinv ? pcg_persist_kernel<T, NCT, K, true> : pcg_persist_kernel<T, NCT, K, false>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, ka, ws, W, spin_limit, staged ? 1u : 0u, hold_wg, hold_us);
    }
    return hipGetLastError();
}

// One block size: the kernels of K = 1, 2 or 3 knots per workgroup
template <typename T, int NN>
static hipError_t launch_persist_n(const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction, uint32_t K)
{
    if (K == 1) return launch_persist_k<T, NN, 1>(a, workspace, s, one_reduction);
    if (K == 2) return launch_persist_k<T, NN, 2>(a, workspace, s, one_reduction);
    return launch_persist_k<T, NN, 3>(a, workspace, s, one_reduction);
}

}  // namespace gbdpcg
