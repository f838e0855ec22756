// bt_sym.hpp -- block-row streaming for SYMMETRIC block-tridiagonal matrices (L_{k+1} = R_k^T).
//
// Only [D_k | R_k] of every block-row is read (2/3 of the bytes); the sub-diagonal contribution
// L_{k+1} x_k is formed from the same registers as R_k^T x_k:
//
//     y_k     +=  D_k x_k + R_k x_{k+1}          (main product, as in bt_device.hpp)
//     y_{k+1} +=  R_k^T x_k                      (transposed product, reduced over the rows)
//
// The reference always reads L_k (utils.cuh:77-83); with L_{k+1} == R_k^T bit for bit the two
// formulations multiply exactly the same numbers, only the summation order differs.  The caller
// opts in (gbdpcg_set_symmetric) or lets the library check the relation on the device first.
//
// Lane map (V = 2 rows per lane, n even, n/2 <= 8): lanes come in aligned groups of 8, lane
// rp = lane & 7 < n/2 owns rows 2rp, 2rp+1, group g = lane >> 3 owns column g + 8s of the 2n-column
// row [D_k | R_k] in step s.  Aligned groups make both reductions VALU-only DPP:
//   * over the rows of a column (transposed product): quad_perm, quad_perm, row_half_mirror
//   * over the 8 column groups (main product): row_ror:8 in the VALU, then two xor-shuffles
// versus four LDS-crossbar shuffle levels per value in the general path.  For n = 14 a step reads
// 8 columns = 448 contiguous bytes with 56 of 64 lanes.
#pragma once

#include "bt_device.hpp"

namespace gbdpcg {

template <typename T, int NCT> struct SymGeom {
    static constexpr uint32_t N_ = NCT > 0 ? NCT : 2;
    static constexpr uint32_t RPC = N_ / 2;                    // live lanes per group
    static constexpr uint32_t COLS = 2 * N_;                   // [D | R]
    static constexpr uint32_t STEPS = (COLS + 7) / 8;
    static constexpr uint32_t REGS = STEPS * 2 * sizeof(T) / 4;  // VGPRs per row per lane
    static constexpr int DEPTH = REGS <= 12 ? 4 : 3;
    static constexpr bool OK = NCT > 0 && NCT % 2 == 0 && RPC <= 8 && RPC >= 4;
};

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 8 lanes of an aligned group; every lane of the group gets the total.
template <typename T> __device__ __forceinline__ T sum_group8(T v)
{
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    return v;
}
// Sum over the lanes with equal (lane & 7), i.e. over the 8 groups of the wave; all lanes get it.
template <typename T> __device__ __forceinline__ T sum_over_groups(T v)
{
    v += dpp_mov<0x128>(v);  // row_ror:8 : the two groups of a 16-lane row
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

template <typename T, int NCT> struct SymCtx {
    using Sg = SymGeom<T, NCT>;
    uint32_t g, rp;
    bool act;
    uint32_t off[Sg::STEPS];   // element offset of this lane's pair inside [D|R], 0 where it owns nothing
    bool val[Sg::STEPS];
    __device__ __forceinline__ explicit SymCtx(uint32_t lane) {
        g = lane >> 3;
        rp = lane & 7u;
        act = rp < Sg::RPC;
#pragma unroll
        for (uint32_t s = 0; s < Sg::STEPS; ++s) {
            const uint32_t c = g + 8 * s;
            val[s] = act && c < Sg::COLS;
            off[s] = val[s] ? c * Sg::N_ + rp * 2 : 0u;
        }
    }
};

// Same prime / run protocol as RowStream (bt_device.hpp): a ring of DEPTH rows in registers,
// branch-free priming, counted waits in the steady state.
template <typename T, int NCT, bool NT = false> struct SymStream {
    using Sg = SymGeom<T, NCT>;
    static constexpr int DEPTH = Sg::DEPTH;
    struct Unit { T a[Sg::STEPS][2]; };
    Unit ring[DEPTH];
    const T *M;  // problem base (block-row k at M + k*3n^2, its [D|R] n^2 further)
    uint32_t k_begin, k_end, k_step, total;

    __device__ __forceinline__ void issue(uint32_t q, int slot, const SymCtx<T, NCT> &cx) {
        const uint32_t k = q < total ? k_begin + q * k_step : 0u;
        const T *base = M + (size_t)k * 3 * Sg::N_ * Sg::N_ + Sg::N_ * Sg::N_;
#pragma unroll
        for (uint32_t s = 0; s < Sg::STEPS; ++s) VecIO<T, 2>::template load<NT>(base + cx.off[s], ring[slot].a[s]);
    }

    __device__ __forceinline__ void prime(const T *__restrict__ M_, uint32_t k_begin_, uint32_t k_end_, uint32_t k_step_,
                                          const SymCtx<T, NCT> &cx) {
        M = M_; k_begin = k_begin_; k_end = k_end_; k_step = k_step_;
        total = k_end > k_begin ? (k_end - k_begin + k_step - 1) / k_step : 0;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            issue((uint32_t)j, j, cx);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Row k: x0 points at x_k (x_{k+1} follows it; n zeros must follow the last knot).
    //   main result  y_k[2rp], y_k[2rp+1]  -> on_main(k, a0, a1)   valid in lanes g == 0 && act
    //   transposed   (R_k^T x_k)[c - n]    -> on_trans(k, c, t)    valid in lanes rp == 0, for the
    //                                         columns c >= n this lane's group owns (k < N-1 only)
    // The kernel is instruction-issue bound (SQ_ACTIVE_INST_ANY ~ kernel cycles per SIMD), so the
    // common case carries no per-element masking: the idle lane of each group (rp = 7 for n = 14) is
    // neutralised once through its x operand, only the ragged last step masks its out-of-row lanes,
    // and the last block-row (whose R block must not be used) takes a separate, masked copy of the loop.
    template <bool LASTROW>
    __device__ __forceinline__ void row_products(const Unit &u, const T *xk, const SymCtx<T, NCT> &cx, T &a0, T &a1,
                                                 T (&t)[Sg::STEPS]) {
        constexpr uint32_t n = Sg::N_;
        using P2 = typename VecOf<T, 2>::type;
        P2 own = *reinterpret_cast<const P2 *>(xk + (cx.act ? cx.rp * 2 : 0u));
        own.x = cx.act ? own.x : T(0);  // idle lanes hold junk matrix elements: their row operand is 0 ...
        own.y = cx.act ? own.y : T(0);
        a0 = T(0);
        a1 = T(0);
#pragma unroll
        for (uint32_t s = 0; s < Sg::STEPS; ++s) {
            const uint32_t c = cx.g + 8 * s;
            T e0 = u.a[s][0], e1 = u.a[s][1];
            constexpr bool kRagged = Sg::COLS % 8 != 0;
            bool keep = true;
            if (kRagged && s == Sg::STEPS - 1) keep = cx.val[s];           // columns past the row
            if (LASTROW && 8 * s + 7 >= n) keep = keep && c < n;           // R_{N-1} is never used (pcg.cuh:106)
            if ((kRagged && s == Sg::STEPS - 1) || (LASTROW && 8 * s + 7 >= n)) {
                e0 = keep ? e0 : T(0);
                e1 = keep ? e1 : T(0);
            }
            const T xv = xk[c < Sg::COLS ? c : Sg::COLS - 1];
            a0 = fma_t(e0, xv, a0);  // ... and their main products are never read
            a1 = fma_t(e1, xv, a1);
            if (8 * s + 7 >= n) {  // this step can touch R columns (compile-time)
                const T tt = fma_t(e1, own.y, e0 * own.x);
                t[s] = (8 * s >= n) ? tt : (c >= n ? tt : T(0));
            } else {
                t[s] = T(0);
            }
        }
    }

    template <typename MainFn, typename TransFn>
    __device__ __forceinline__ void consume(uint32_t q, int slot, const T *x, uint32_t N, const SymCtx<T, NCT> &cx,
                                            bool refill, MainFn &&on_main, TransFn &&on_trans) {
        constexpr uint32_t n = Sg::N_;
        const uint32_t k = k_begin + q * k_step;
        const T *xk = x + k * n;
        const bool lastrow = k == N - 1;  // wave-uniform
        T a0, a1, t[Sg::STEPS];
        if (lastrow) row_products<true>(ring[slot], xk, cx, a0, a1, t);
        else row_products<false>(ring[slot], xk, cx, a0, a1, t);
        if (refill) issue(q + DEPTH, slot, cx);
#pragma unroll
        for (uint32_t s = 0; s < Sg::STEPS; ++s) {
            if (8 * s + 7 >= n) t[s] = sum_group8(t[s]);
        }
        a0 = sum_over_groups(a0);
        a1 = sum_over_groups(a1);
        on_main(k, a0, a1);
        if (!lastrow) {
#pragma unroll
            for (uint32_t s = 0; s < Sg::STEPS; ++s) {
                const uint32_t c = cx.g + 8 * s;
                if (8 * s + 7 >= n && c >= n && c < Sg::COLS) on_trans(k, c, t[s]);
            }
        }
    }

    template <typename MainFn, typename TransFn>
    __device__ __forceinline__ void run(const T *x, uint32_t N, const SymCtx<T, NCT> &cx, MainFn &&on_main,
                                        TransFn &&on_trans) {
        uint32_t q0 = 0;
        for (; q0 + 2 * DEPTH <= total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) consume(q0 + j, j, x, N, cx, true, on_main, on_trans);
        }
        for (; q0 < total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const uint32_t q = q0 + j;
                if (q < total) consume(q, j, x, N, cx, q + DEPTH < total, on_main, on_trans);
            }
        }
    }
};

}  // namespace gbdpcg
