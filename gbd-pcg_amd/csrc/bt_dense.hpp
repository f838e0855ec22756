// bt_dense.hpp -- the register-resident ("dense") lane map shared by pcg_resident.hip (one workgroup per problem) and
// pcg_cluster.hip (a problem over a few workgroups): a lane owns V whole rows of one block-row of [L|D|R].
#pragma once

#include "bt_device.hpp"

namespace gbdpcg {

// When both matrices of a problem fit the register file of ONE 8-wave workgroup they are read from
// HBM exactly once and an iteration touches only LDS.  This is the reference's own idea (its
// block-rows sit in shared memory for the whole solve, pcg.cuh:104-110) moved one level up the
// hierarchy, with the whole problem inside one workgroup so that no grid barrier exists.
//
// Lane map (different from the streaming one, where every instruction must be a dense global read):
// a lane owns V whole ROWS of one block-row -- all 3n columns, 3n*V registers per matrix -- so a
// block-row product is 3n FMAs per row with NO cross-lane fold, columns accumulated in ascending
// order exactly like bdmv (utils.cuh:77-81).  n/V lanes make a block-row, BPW = 64/(n/V) block-rows
// make a wave (n=14, V=2: 7 lanes x 9 block-rows = 63 lanes; the reference keeps n of 64 threads
// busy), 8 waves cover N <= 8*BPW knots.  x is read from LDS as 8/16-byte pairs shared by the lanes
// of a block-row.  Edge columns (L_0, R_{N-1}) and rows past N are zeroed once, at load time.
template <typename T, int NCT, int V> struct DenseGeom {
    static constexpr uint32_t N_ = NCT > 0 ? NCT : 2;
    static constexpr uint32_t LPB = N_ / V > 0 ? N_ / V : 1;  // lanes per block-row
    static constexpr uint32_t BPW = kWave / LPB;        // block-rows per wave
    static constexpr uint32_t COLS = 3 * N_;
    static constexpr uint32_t REGS = COLS * V * sizeof(T) / 4;  // VGPRs per lane per matrix
    static constexpr uint32_t WAVES = 8;
    static constexpr uint32_t MAX_KNOTS = WAVES * BPW;
};

template <typename T, int NCT, int V> struct DenseTile {
    T a[DenseGeom<T, NCT, V>::COLS][V];
};

template <typename T, int NCT, int V> struct DenseCtx {
    uint32_t k;       // this lane's block-row (knot index in the problem)
    uint32_t kl;      // ... counted from the first knot of the workgroup (== k when the workgroup has the whole problem)
    uint32_t rp;      // row group inside it
    bool live;        // lane maps to a real row of a real block-row
    // the workgroup holds knots [k_lo, k_lo + cnt)
    __device__ __forceinline__ DenseCtx(uint32_t wave, uint32_t lane, uint32_t cnt, uint32_t k_lo = 0) {
        using Dg = DenseGeom<T, NCT, V>;
        const uint32_t b = lane / Dg::LPB;
        rp = lane - b * Dg::LPB;
        kl = wave * Dg::BPW + b;
        k = k_lo + kl;
        live = b < Dg::BPW && kl < cnt;
    }
};

// (C0, C1: the columns [C0, C1) of the block-row only -- pcg_cluster.hip brings a tile in in two parts where half of it goes on to LDS)
template <typename T, int NCT, int V, uint32_t C0 = 0, uint32_t C1 = 3 * (NCT > 0 ? NCT : 2)>
__device__ __forceinline__ void dense_load(const T *__restrict__ M, uint32_t N, const DenseCtx<T, NCT, V> &dc,
                                           DenseTile<T, NCT, V> &tl)
{
    using Dg = DenseGeom<T, NCT, V>;
    const uint32_t k = dc.live ? dc.k : 0u;
    const T *src = M + (size_t)k * 3 * Dg::N_ * Dg::N_ + dc.rp * V;
    const uint32_t c_lo = dc.k == 0 ? Dg::N_ : 0u, c_hi = dc.k == N - 1 ? 2 * Dg::N_ : 3 * Dg::N_;
#pragma unroll
    for (uint32_t c = C0; c < C1; ++c) {
        T v[V];
        VecIO<T, V>::load(src + c * Dg::N_, v);
        const bool keep = dc.live && c >= c_lo && c < c_hi;
#pragma unroll
        for (int j = 0; j < V; ++j) tl.a[c][j] = keep ? v[j] : T(0);
    }
}

// y_k = [L|D|R]_k * X-window for this lane's rows, left in registers.  X: the operand with one knot of padding in
// front of the workgroup's first knot (zeros, or the neighbouring workgroup's boundary knot).
// CHAINS = 1: one accumulator per row, columns in ascending order exactly like bdmv (utils.cuh:77-81).  CHAINS = 3: one
// accumulator per block, y = (L x_{k-1} + D x_k) + R x_{k+1}, each block's columns ascending: three independent chains of
// 14 packed fma's instead of one of 42 (a dependent v_pk_fma_f32 issues every ~10 cycles: the single chain cost
// ~1,000 cycles per product with two waves per SIMD, measured in pcg_cluster.hip).
// TAIL > 0 (CHAINS = 3 only): the last TAIL columns of the block-row are not in the tile but in LDS, `tail[t * tstride]` =
// this lane's two rows of column COLS - TAIL + t (a deliberate, cheap spill: pcg_cluster.hip keeps 8 of its 168 matrix
// registers there, because hipcc otherwise spills a few of them to scratch and reloads them inside every product).
// A lane's V rows of one tail column as they lie in LDS: a pair (V = 2) or a single value (V = 1).
template <typename T, int V> struct DenseTailElem { using type = typename VecOf<T, 2>::type; };
template <typename T> struct DenseTailElem<T, 1> { using type = T; };
__device__ __forceinline__ float dense_tail_row(const float &v, int) { return v; }
__device__ __forceinline__ double dense_tail_row(const double &v, int) { return v; }
__device__ __forceinline__ float dense_tail_row(const float2 &v, int j) { return j == 0 ? v.x : v.y; }
__device__ __forceinline__ double dense_tail_row(const double2 &v, int j) { return j == 0 ? v.x : v.y; }

#ifndef GBDPCG_DENSE_AH
#define GBDPCG_DENSE_AH 2   // steps ahead of its use an operand pair of the three-chain product is requested (1: round 2; 2 measured 1-1.6 % faster on the cluster shapes and 12 registers lighter at n = 14; 3: no further gain)
#endif
template <typename T, int NCT, int V, int CHAINS = 1, int TAIL = 0>
__device__ __forceinline__ void dense_mv(const DenseTile<T, NCT, V> &tl, const T *X, const DenseCtx<T, NCT, V> &dc,
                                         T (&acc)[V], const typename DenseTailElem<T, V>::type *tail = nullptr, uint32_t tstride = 0,
                                         const typename DenseTailElem<T, V>::type *tail_d = nullptr)
{
    using Dg = DenseGeom<T, NCT, V>;
    using P2 = typename VecOf<T, 2>::type;
    using TV = typename DenseTailElem<T, V>::type;
    static_assert(CHAINS == 1 || (CHAINS == 3 && Dg::N_ % 2 == 0), "one chain, or one per block (whole x pairs per block)");
    static_assert(TAIL == 0 || (CHAINS == 3 && V <= 2 && TAIL % 2 == 0 && TAIL <= 2 * (int)Dg::N_), "tail columns: whole pairs of the last block (and of the one before it)");
    // tail columns of the last block (R) and of the middle one (D): TAIL > n puts all of R and the last TAIL - n columns of D in LDS,
    // the D part behind its own pointer (`tail_d`, same stride)
    constexpr int TAIL_R = TAIL < (int)Dg::N_ ? TAIL : (int)Dg::N_, TAIL_D = TAIL - TAIL_R;
    const uint32_t kl = dc.live ? dc.kl : 0u;
    const P2 *xk = reinterpret_cast<const P2 *>(X + kl * Dg::N_);  // column c of local row kl multiplies X[kl*n + c]
    T part[CHAINS][V];
#pragma unroll
    for (int q = 0; q < CHAINS; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) part[q][j] = T(0);
    if constexpr (CHAINS == 1 && Dg::COLS % 2 != 0) {
        // odd block size: the operand pairs of a knot are not 8-byte aligned in the window; one entry per read
        const T *x1 = X + kl * Dg::N_;
#pragma unroll
        for (uint32_t c = 0; c < Dg::COLS; ++c) {
            const T xv = x1[c];
#pragma unroll
            for (int j = 0; j < V; ++j) part[0][j] = fma_t(tl.a[c][j], xv, part[0][j]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = part[0][j];
    } else if constexpr (CHAINS == 1) {
#pragma unroll
        for (uint32_t c = 0; c < Dg::COLS; c += 2) {
            const P2 xv = xk[c / 2];
#pragma unroll
            for (int j = 0; j < V; ++j) part[0][j] = fma_t(tl.a[c][j], xv.x, part[0][j]);
#pragma unroll
            for (int j = 0; j < V; ++j) part[0][j] = fma_t(tl.a[c + 1][j], xv.y, part[0][j]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = part[0][j];
    } else {
        // The three blocks side by side: independent work in every step.  The steps are pinned in source order with the
        // operands of the next one requested before the fma's of this one: left to itself hipcc hoists all 21 operand reads
        // to the top (42 more live registers next to 168 of matrix data) and spills elsewhere in the kernel.
        // AH steps ahead (GBDPCG_DENSE_AH): an LDS read issued one step before its use is not back in time -- a step is six packed
        // fmas per wave -- and every step then stalls on it (measured in pcg_cluster.hip: 950 cycles per product, 7 steps).
        constexpr uint32_t STEPS = Dg::N_ / 2, AH = GBDPCG_DENSE_AH < STEPS ? GBDPCG_DENSE_AH : STEPS;
        P2 xq[STEPS][3];
        TV tq[STEPS][2], td[STEPS][2];
        auto request = [&](uint32_t st) __attribute__((always_inline)) {
            const uint32_t c = 2 * st;
#pragma unroll
            for (int q = 0; q < 3; ++q) xq[st][q] = xk[(q * Dg::N_ + c) / 2];
            if constexpr (TAIL > 0) {
                if (c >= Dg::N_ - TAIL_R) {   // this step's two columns of the last block come from LDS
                    tq[st][0] = tail[(c - (Dg::N_ - TAIL_R)) * tstride];
                    tq[st][1] = tail[(c + 1 - (Dg::N_ - TAIL_R)) * tstride];
                }
            }
            if constexpr (TAIL_D > 0) {
                if (c >= Dg::N_ - TAIL_D) {   // ... and of the middle one
                    td[st][0] = tail_d[(c - (Dg::N_ - TAIL_D)) * tstride];
                    td[st][1] = tail_d[(c + 1 - (Dg::N_ - TAIL_D)) * tstride];
                }
            }
        };
#pragma unroll
        for (uint32_t st = 0; st < AH; ++st) request(st);
#pragma unroll
        for (uint32_t st = 0; st < STEPS; ++st) {
            const uint32_t c = 2 * st;
            if (st + AH < STEPS) request(st + AH);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool from_r = TAIL > 0 && q == 2 && c >= Dg::N_ - TAIL_R, from_d = TAIL_D > 0 && q == 1 && c >= Dg::N_ - TAIL_D;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const T m = from_r ? dense_tail_row(tq[st][0], j) : from_d ? dense_tail_row(td[st][0], j) : tl.a[q * Dg::N_ + c][j];
                    part[q][j] = fma_t(m, xq[st][q].x, part[q][j]);
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const T m = from_r ? dense_tail_row(tq[st][1], j) : from_d ? dense_tail_row(td[st][1], j) : tl.a[q * Dg::N_ + c + 1][j];
                    part[q][j] = fma_t(m, xq[st][q].y, part[q][j]);
                }
            }
            if constexpr (TAIL > 0 || sizeof(T) == 8 || (GBDPCG_DENSE_AH > 1)) {
                // with tail operands coming out of LDS (and in fp64, where everything is twice as wide, and with operands requested
                // more than a step ahead) the accumulators are anchored here as well: the sched_barrier alone orders the LDS reads,
                // and hipcc then sinks every fma of the product below the last of them (all operands live at once: 80
                // registers, and matrix registers spilled inside the iteration)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int j = 0; j < V; ++j) asm volatile("" : "+v"(part[q][j]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = (part[0][j] + part[1][j]) + part[2][j];
    }
}

// Staged tile loads (a lane's V rows of a column are 8 bytes: fp32 with V = 2 or fp64 with V = 1; whole 16-byte pieces per block).
// A wave's knots are contiguous in memory; one block (L, D or R: n^2 elements) of each of its up to BPW knots is fetched per
// stage with LOADS LDS-DMA instructions (64 lanes x 16 bytes, dense and coalesced, no VGPR in between; n = 14, fp32: 9 knots x
// 49 pieces, 7 instructions) into the wave's own staging buffer, and the lanes then pick their rows up with 8-byte LDS reads.
// The direct form (dense_load above) reads 8 bytes per lane at an n-element stride: 6 n sparse instructions per matrix, bound
// by the address path -- in pcg_cluster.hip 28 us of a 40 us round went there.
template <typename T, int NCT, int V> struct DenseStage {
    using Dg = DenseGeom<T, NCT, V>;
    static constexpr uint32_t PIECES = NCT * NCT * sizeof(T) / 16;               // 16-byte pieces per n x n block
    static constexpr uint32_t LOADS = (Dg::BPW * PIECES + 63) / 64;              // LDS-DMA instructions per stage
    // (the stages are written out for 4 .. 9 loads: block sizes below 8 take the direct loads)
    static constexpr bool OK = V * sizeof(T) == 8 && (NCT * NCT * sizeof(T)) % 16 == 0 && LOADS >= 4 && LOADS <= 9;
    static constexpr uint32_t BYTES = LOADS * 64 * 16;                           // one staging buffer of one wave
    static constexpr uint32_t BLOCK_ROW_BYTES = 3 * NCT * NCT * sizeof(T);
};

// Issue one stage: lane l of instruction i moves the 16-byte piece q = 64 i + l of the wave's `nk` blocks (PIECES each, blocks
// one block-row apart) from base to lds_addr + 16 q.  Lanes beyond the last piece re-read piece 0 into slots nobody picks
// up: every lane of every instruction is live, so a stage is always exactly LOADS loads on the wave's counter.  The offsets
// are recomputed from the lane number at every issue (kept in registers across the stages they were spilled, and every
// reload from scratch came with an s_waitcnt vmcnt(0) that drained the stages in flight).  The loads are written in asm
// (M0 carries the LDS address) and so are invisible to hipcc's counters: the caller waits with dense_stage_wait before
// it reads the buffer, and never has more than two stages in flight.
template <typename T, int NCT, int V>
__device__ __forceinline__ void dense_stage_issue(const T *base, uint32_t lane, uint32_t nk, uint32_t lds_addr)
{
    using St = DenseStage<T, NCT, V>;
    static_assert(St::OK && St::LOADS >= 4 && St::LOADS <= 9, "stage sizes written out: 4 .. 9 loads");
    uint32_t lo = lane;
    asm volatile("" : "+v"(lo));
    uint32_t rel[St::LOADS];
#pragma unroll
    for (uint32_t i = 0; i < St::LOADS; ++i) {
        const uint32_t q = i * 64 + lo, j = q / St::PIECES;   // (a constant divisor: multiply and shift)
        rel[i] = j < nk ? q * 16 + j * (St::BLOCK_ROW_BYTES - St::PIECES * 16) : 0u;
    }
    unsigned keep;
    // ONE asm statement per stage: M0 must not be touched by anything the compiler schedules in between
    if constexpr (St::LOADS == 4) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "s"(lds_addr)
                     : "memory", "scc");
    }
    else if constexpr (St::LOADS == 5) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "v"(rel[4]), "s"(lds_addr)
                     : "memory", "scc");
    }
    else if constexpr (St::LOADS == 6) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "v"(rel[4]), "v"(rel[5]), "s"(lds_addr)
                     : "memory", "scc");
    }
    else if constexpr (St::LOADS == 7) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "v"(rel[4]), "v"(rel[5]), "v"(rel[6]), "s"(lds_addr)
                     : "memory", "scc");
    }
    else if constexpr (St::LOADS == 8) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "v"(rel[4]), "v"(rel[5]), "v"(rel[6]), "v"(rel[7]), "s"(lds_addr)
                     : "memory", "scc");
    }
    else if constexpr (St::LOADS == 9) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"   // the reads of the stage that used this buffer are done
                     "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(rel[0]), "v"(rel[1]), "v"(rel[2]), "v"(rel[3]), "v"(rel[4]), "v"(rel[5]), "v"(rel[6]), "v"(rel[7]), "v"(rel[8]),
                       "s"(lds_addr)
                     : "memory", "scc");
    }
}
// all but the youngest `newer` stages (LOADS loads each) of this wave have landed
template <typename T, int NCT, int V, int NEWER> __device__ __forceinline__ void dense_stage_wait()
{
    if constexpr (NEWER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DenseStage<T, NCT, V>::LOADS) : "memory");
}

typedef float2 __attribute__((may_alias)) dense_float2_alias;

// This lane's V rows (8 bytes per column) of block BLK (0 = L, 1 = D, 2 = R) out of the wave's staging buffer into its tile.
template <typename T, int NCT, int V, int BLK>
__device__ __forceinline__ void dense_stage_pick(const unsigned char *buf, const DenseCtx<T, NCT, V> &dc, uint32_t b9, uint32_t N,
                                                   DenseTile<T, NCT, V> &tl)
{
    constexpr uint32_t n = NCT;
    // L_0 and R_{N-1} are never used (pcg.cuh:105-106); lanes without a row hold zeros
    const bool keep = dc.live && !(BLK == 0 && dc.k == 0) && !(BLK == 2 && dc.k == N - 1);
    const dense_float2_alias *src = reinterpret_cast<const dense_float2_alias *>(buf + (size_t)b9 * n * n * sizeof(T) + dc.rp * 8);
#pragma unroll
    for (uint32_t c = 0; c < n; ++c) {
        const float2 v = src[c * n * sizeof(T) / 8];
        if constexpr (sizeof(T) == 4) {
            tl.a[BLK * n + c][0] = keep ? v.x : 0.f;
            tl.a[BLK * n + c][1] = keep ? v.y : 0.f;
        } else {
            const unsigned long long b = ((unsigned long long)__builtin_bit_cast(uint32_t, v.y) << 32) | __builtin_bit_cast(uint32_t, v.x);
            tl.a[BLK * n + c][0] = keep ? __builtin_bit_cast(double, b) : 0.0;
        }
    }
}

template <typename T, int NCT, int V> constexpr size_t dense_stage_lds_bytes() { return (size_t)2 * DenseGeom<T, NCT, V>::WAVES * DenseStage<T, NCT, V>::BYTES; }

// Both tiles of a workgroup's knots [k_lo, k_lo + cnt) through the staging buffers (`stage`: dense_stage_lds_bytes of LDS, 16-byte
// aligned; matrices 16-byte aligned): six stages (S: L D R, Pinv: L D R; three when P == nullptr), two in flight, alternating
// buffers.  `between` is called once, behind the first two stage issues: loads requested there (vectors) are younger than
// two stages, which makes the first wait stricter than it has to be, never laxer.  tP is defined on every path (zeros
// without a preconditioner): a conditionally loaded tile is carried around the caller's problem loop by hipcc, all of it.
// P_TAIL (0, n or 2n; 2n: the D block as well, re-laid inside the first buffer, the R block inside the second -- dense_mv's tail_d
// and tail): the R block of Pinv is not picked into tP but left in LDS for dense_mv's TAIL form -- in the wave's OWN
// first staging buffer (the last stage arrives in the second one, and a wave's LDS operations execute in order), one
// 8-byte element per lane and column, lane-contiguous: tail[c * 64 + lane].  The wave must not stage anything else before
// its solve is over.
template <typename T, int NCT, int V, int P_TAIL = 0, typename Between>
__device__ __forceinline__ void dense_staged_load(const T *S, const T *P, uint32_t N, const DenseCtx<T, NCT, V> &dc,
                                                  uint32_t wave, uint32_t lane, uint32_t k_lo, uint32_t cnt, unsigned char *stage,
                                                  DenseTile<T, NCT, V> &tS, DenseTile<T, NCT, V> &tP, Between between)
{
    static_assert(P_TAIL == 0 || ((P_TAIL == NCT || P_TAIL == 2 * NCT) && (size_t)NCT * 64 * 8 <= DenseStage<T, NCT, V>::BYTES),
                  "the whole R block (or the D and the R block), each inside one staging buffer");
    using Dg = DenseGeom<T, NCT, V>;
    constexpr uint32_t n = NCT;
    // this wave's knots [kw, kw + nk)
    const uint32_t kw = k_lo + wave * Dg::BPW;
    const uint32_t nk = wave * Dg::BPW < cnt ? (cnt - wave * Dg::BPW < Dg::BPW ? cnt - wave * Dg::BPW : Dg::BPW) : 0u;
    const uint32_t kbase = kw < N ? kw : N - 1;
    unsigned char *buf0 = stage + wave * DenseStage<T, NCT, V>::BYTES;
    unsigned char *buf1 = buf0 + Dg::WAVES * DenseStage<T, NCT, V>::BYTES;
    const uint32_t lds0 = (uint32_t)(uintptr_t)buf0, lds1 = (uint32_t)(uintptr_t)buf1;
    const T *Sw = S + (size_t)kbase * 3 * n * n, *Pw = (P ? P : S) + (size_t)kbase * 3 * n * n;
    const uint32_t b9 = dc.live ? lane / Dg::LPB : 0u;
    dense_stage_issue<T, NCT, V>(Sw, lane, nk, lds0);
    dense_stage_issue<T, NCT, V>(Sw + n * n, lane, nk, lds1);
    between();
    dense_stage_wait<T, NCT, V, 1>();
    dense_stage_pick<T, NCT, V, 0>(buf0, dc, b9, N, tS);
    dense_stage_issue<T, NCT, V>(Sw + 2 * n * n, lane, nk, lds0);
    dense_stage_wait<T, NCT, V, 1>();
    dense_stage_pick<T, NCT, V, 1>(buf1, dc, b9, N, tS);
    if (P) dense_stage_issue<T, NCT, V>(Pw, lane, nk, lds1);
    if (P) dense_stage_wait<T, NCT, V, 1>(); else dense_stage_wait<T, NCT, V, 0>();
    dense_stage_pick<T, NCT, V, 2>(buf0, dc, b9, N, tS);
    if (P) {
        dense_stage_issue<T, NCT, V>(Pw + n * n, lane, nk, lds0);
        dense_stage_wait<T, NCT, V, 1>();
        dense_stage_pick<T, NCT, V, 0>(buf1, dc, b9, N, tP);
        dense_stage_issue<T, NCT, V>(Pw + 2 * n * n, lane, nk, lds1);
        dense_stage_wait<T, NCT, V, 1>();
        // One block from the staging layout to the tail layout (tail[c * 64 + lane], one 8-byte element per lane and column) INSIDE
        // its buffer: every read of the wave is issued before its first write, and a wave's LDS operations execute in order.
        auto in_place = [&](unsigned char *buf, bool keep) __attribute__((always_inline)) {
            const dense_float2_alias *src = reinterpret_cast<const dense_float2_alias *>(buf + (size_t)b9 * n * n * sizeof(T) + dc.rp * 8);
            dense_float2_alias *tail = reinterpret_cast<dense_float2_alias *>(buf) + lane;
            float2 v[n];
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t c = 0; c < n; ++c) v[c] = src[c * n * sizeof(T) / 8];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (uint32_t c = 0; c < n; ++c) tail[c * 64] = keep ? v[c] : make_float2(0.f, 0.f);
            asm volatile("" ::: "memory");
        };
        if constexpr (P_TAIL == 2 * NCT) in_place(buf0, dc.live);
        else dense_stage_pick<T, NCT, V, 1>(buf0, dc, b9, N, tP);
        dense_stage_wait<T, NCT, V, 0>();
        if constexpr (P_TAIL == 0) {
            dense_stage_pick<T, NCT, V, 2>(buf1, dc, b9, N, tP);
        } else if constexpr (P_TAIL == 2 * NCT) {
            in_place(buf1, dc.live && dc.k != N - 1);   // R_{N-1} is never used; lanes without a row hold zeros
        } else {
            const bool keep = dc.live && dc.k != N - 1;   // R_{N-1} is never used; lanes without a row hold zeros
            const dense_float2_alias *src = reinterpret_cast<const dense_float2_alias *>(buf1 + (size_t)b9 * n * n * sizeof(T) + dc.rp * 8);
            dense_float2_alias *tail = reinterpret_cast<dense_float2_alias *>(buf0) + lane;
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t c = 0; c < n; ++c) {
                const float2 v = src[c * n * sizeof(T) / 8];
                tail[c * 64] = keep ? v : make_float2(0.f, 0.f);
            }
            asm volatile("" ::: "memory");
        }
    } else {
#pragma unroll
        for (uint32_t cc = 0; cc < Dg::COLS; ++cc)
#pragma unroll
            for (int j = 0; j < V; ++j) tP.a[cc][j] = T(0);
    }
}

}  // namespace gbdpcg
