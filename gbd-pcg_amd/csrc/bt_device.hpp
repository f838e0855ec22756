// bt_device.hpp -- wavefront-level primitives for block-tridiagonal [L|D|R] matrices (gfx950).
//
// What replaces what (reference paths relative to /root/reference):
//   RowStream      <- loadbdVec + bdmv              (include/utils.cuh:9-85)
//   wave_sum       <- glass::dot / glass::reduce    (call sites include/pcg.cuh:144-149,163-169,187-193)
//
// Design (not a translation).  The reference gives one CUDA block to a knot and lets thread r
// walk the 3n columns of row r serially (n of 64 threads busy, smem-resident matrix).  Here a
// block-row is streamed straight from HBM by ONE wavefront, flattened:
//
//   block-row k = n x 3n column-major = 3n^2 contiguous elements
//   lane l  ->  (rp, g) = (l % (n/V), l / (n/V)):  V consecutive rows rp*V.. of column-group g
//   step s  ->  the wave reads the G = floor(64 / (n/V)) columns  s*G .. s*G+G-1  as ONE
//               contiguous chunk of G*n elements; lane l reads elements [l*V, l*V+V) of it
//
// so every load instruction is a dense, ascending V*sizeof(T)-byte-per-lane access (8 B for
// n = 14 fp32 with 63/64 lanes live -- measured at the read ceiling of the chip, see
// profiles/r01_bw_probe_*.txt; 16 B for n = 36 fp64), each lane keeps ONE fixed row set and
// accumulates over columns in registers, and only a log2(G) shuffle tree is needed per
// block-row.  The x operand ([x_{k-1}; x_k; x_{k+1}], 3n values) is read from LDS as G
// broadcast addresses per step.  Loads are software-pipelined through a register ring of
// DEPTH units (a unit = a group of steps) so a wave always has DEPTH-1 units in flight while
// it multiplies and reduces the oldest one.  MFMA is not used: the contraction is a GEMV with arithmetic intensity
// ~0.5 flop/B, bounded by HBM.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gbdpcg {

constexpr uint32_t kWave = 64;

// One correctly-rounded fused multiply-add in T's own precision.  (__builtin_fma is the DOUBLE
// builtin: given floats it converts, does an fp64 FMA and rounds back -- slow and not fp32 math.)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int V> struct VecOf;
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<double, 1> { using type = double; };
template <> struct VecOf<double, 2> { using type = double2; };

// NT = non-temporal cache policy (global_load ... nt): for data read once per launch.  Measured on
// Infinity-Cache-cold matrices (tools/bw_probe.hip) the same access shape streams 8 % faster with it.
template <typename T, int V> struct NtVec { typedef T type __attribute__((ext_vector_type(V))); };
template <typename T, int V> struct VecIO;
template <typename T> struct VecIO<T, 1> {
    template <bool NT = false> static __device__ __forceinline__ void load(const T *p, T (&a)[1]) {
        a[0] = NT ? __builtin_nontemporal_load(p) : *p;
    }
};
template <typename T> struct VecIO<T, 2> {
    template <bool NT = false> static __device__ __forceinline__ void load(const T *p, T (&a)[2]) {
        if constexpr (NT) {
            const auto v = __builtin_nontemporal_load(reinterpret_cast<const typename NtVec<T, 2>::type *>(p));
            a[0] = v.x; a[1] = v.y;
        } else {
            using VT = typename VecOf<T, 2>::type;
            VT v = *reinterpret_cast<const VT *>(p);
            a[0] = v.x; a[1] = v.y;
        }
    }
};
template <typename T> struct VecIO<T, 4> {
    template <bool NT = false> static __device__ __forceinline__ void load(const T *p, T (&a)[4]) {
        if constexpr (NT) {
            const auto v = __builtin_nontemporal_load(reinterpret_cast<const typename NtVec<T, 4>::type *>(p));
            a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        } else {
            using VT = typename VecOf<T, 4>::type;
            VT v = *reinterpret_cast<const VT *>(p);
            a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        }
    }
};

__host__ __device__ constexpr uint32_t pow2_ceil(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// Lane -> (row-group, column-group) map of one wavefront for block size n, V rows per lane.
// NCT > 0 makes everything a compile-time constant.
template <int NCT, int V> struct LaneMap {
    uint32_t n;      // block size
    uint32_t rpc;    // row-groups per column = n / V            (requires rpc <= 64)
    uint32_t G;      // columns per step = 64 / rpc
    uint32_t steps;  // ceil(3n / G)
    uint32_t rp, g;  // this lane's row-group / column-group
    bool active;     // lane < G * rpc

    __device__ __forceinline__ explicit LaneMap(uint32_t n_rt, uint32_t lane) {
        n = NCT ? (uint32_t)NCT : n_rt;
        rpc = n / V;
        G = kWave / rpc;
        steps = (3 * n + G - 1) / G;
        g = lane / rpc;
        rp = lane - g * rpc;
        active = lane < G * rpc;
    }
};

// Compile-time geometry of the pipelined stream (NCT > 0; for NCT == 0 these are placeholders
// and the runtime geometry lives in StreamCtx<T, 0, V>).
template <typename T, int NCT, int V> struct StreamGeom {
    static constexpr uint32_t N_ = NCT > 0 ? NCT : 1;
    static constexpr uint32_t RPC = N_ / V > 0 ? N_ / V : 1;
    static constexpr uint32_t G = kWave / RPC > 0 ? kWave / RPC : 1;
    static constexpr uint32_t STEPS = (3 * N_ + G - 1) / G;
    // a unit = CH steps held in registers at once; keep a unit at <= ~24 VGPRs per lane
    static constexpr uint32_t REGS_PER_STEP = V * sizeof(T) / 4;
    static constexpr uint32_t CH_MAX = 24 / REGS_PER_STEP > 0 ? 24 / REGS_PER_STEP : 1;
    static constexpr uint32_t UPR = (STEPS + CH_MAX - 1) / CH_MAX;  // units per row
    static constexpr uint32_t CH = (STEPS + UPR - 1) / UPR;         // steps per unit (balanced)
    // ring depth: ~40-72 VGPRs of matrix data in flight per lane
    static constexpr int DEPTH = CH * REGS_PER_STEP <= 12 ? 4 : 3;
};

template <typename T, int NCT, int V> struct StreamUnit {
    T a[StreamGeom<T, NCT, V>::CH][V];
};

// Per-lane constants of the pipelined stream: two element offsets and two x columns, so that
// every load of a unit is `scalar row base + one of two VGPR offsets + immediate`.
//   * lanes outside the map (lane >= G*rpc) read element 0 of the step instead of running past it
//   * on the ragged last step (3n % G != 0) lanes whose column would leave the block-row read
//     element 0 as well and are zeroed in fma_unit
// Lanes outside the map compute junk that fold_groups never picks up.
template <typename T, int NCT, int V> struct StreamCtx {
    using Gm = StreamGeom<T, NCT, V>;
    static constexpr bool RAGGED = (3 * Gm::N_) % Gm::G != 0;
    uint32_t off_lane, off_last;  // element offsets inside a step chunk
    uint32_t xcol, xcol_last;     // x column read at step 0 / at the last step (clamped into the row)
    bool ok_last;                 // this lane's column of the last step is inside the block-row
    __device__ __forceinline__ StreamCtx(const LaneMap<NCT, V> &m, uint32_t lane) {
        const uint32_t c_last = m.g + Gm::G * (Gm::STEPS - 1);
        ok_last = m.active && c_last < 3 * Gm::N_;
        off_lane = m.active ? lane * V : 0u;
        off_last = ok_last ? lane * V : 0u;
        xcol = m.active ? m.g : 0u;
        xcol_last = ok_last ? c_last : 3 * Gm::N_ - 1;
    }
};

// Issue the loads of unit u of the block-row at Mk: CH back-to-back load instructions, no branches.
template <typename T, int NCT, int V, bool NT = false>
__device__ __forceinline__ void load_unit(const T *__restrict__ Mk, uint32_t u, const StreamCtx<T, NCT, V> &cx,
                                          StreamUnit<T, NCT, V> &t)
{
    using Gm = StreamGeom<T, NCT, V>;
    // The two offsets as values of their own: where the unit that holds the last step is only known at run time (n = 24 with
    // V = 4: two units per row), hipcc turned the select between the two ADJACENT members into an indexed load, and with it the
    // whole stream object -- the register ring included -- into scratch (552 bytes per lane, 2.8 TB/s instead of 5.6).
    uint32_t o_lane = cx.off_lane, o_last = cx.off_last;
    if (StreamCtx<T, NCT, V>::RAGGED) asm volatile("" : "+v"(o_last));
#pragma unroll
    for (uint32_t j = 0; j < Gm::CH; ++j) {
        const uint32_t s = u * Gm::CH + j;
        if (Gm::STEPS % Gm::CH == 0 || s < Gm::STEPS) {
            const uint32_t off = (StreamCtx<T, NCT, V>::RAGGED && s == Gm::STEPS - 1) ? o_last : o_lane;
            VecIO<T, V>::template load<NT>(Mk + s * (Gm::G * Gm::N_) + off, t.a[j]);
        }
    }
}

// acc += unit * x.  xg / xl point at this lane's x entries of the row: xg = &x_row[xcol] (then
// column group s sits G*s further), xl = &x_row[xcol_last] (the ragged last step).  Callers build
// them as `lane base + row offset` so that rows share address registers.
// EDGE rows (k = 0 or k = N-1) additionally zero the columns outside [c_lo, c_hi): L_0 / R_{N-1}
// may hold anything and must not reach the result (pcg.cuh:105-106, utils.cuh:58-75).
template <typename T, int NCT, int V, bool EDGE>
__device__ __forceinline__ void fma_unit(const StreamUnit<T, NCT, V> &t, uint32_t u, const T *xg, const T *xl,
                                         const StreamCtx<T, NCT, V> &cx, uint32_t g, uint32_t c_lo, uint32_t c_hi,
                                         T (&acc)[V])
{
    using Gm = StreamGeom<T, NCT, V>;
#pragma unroll
    for (uint32_t j = 0; j < Gm::CH; ++j) {
        const uint32_t s = u * Gm::CH + j;
        if (Gm::STEPS % Gm::CH == 0 || s < Gm::STEPS) {
            const bool last = StreamCtx<T, NCT, V>::RAGGED && s == Gm::STEPS - 1;
            const T xv = last ? xl[0] : xg[Gm::G * s];
            bool keep = last ? cx.ok_last : true;
            if (EDGE) {
                const uint32_t c = g + Gm::G * s;
                keep = keep && c >= c_lo && c < c_hi;
            }
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fma_t((EDGE || last) ? (keep ? t.a[j][v] : T(0)) : t.a[j][v], xv, acc[v]);
        }
    }
}

// Fold the G column-groups: halving tree over g with a lane stride of rpc.  Afterwards lanes
// with g == 0 (and active) hold the row values.
template <typename T, int NCT, int V>
__device__ __forceinline__ void fold_groups(const LaneMap<NCT, V> &m, T (&acc)[V])
{
    uint32_t size = m.G;
    for (uint32_t off = pow2_ceil(m.G) >> 1; off >= 1; off >>= 1) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const T other = __shfl_down(acc[v], off * m.rpc, kWave);
            if (m.g < off && m.g + off < size) acc[v] += other;
        }
        size = off;
    }
}

// The same fold for R independent rows at once: each tree level issues all R*V shuffles before any
// add, so the LDS-crossbar latency of a level is paid once, not R times.
template <typename T, int NCT, int V, int R>
__device__ __forceinline__ void fold_groups_multi(const LaneMap<NCT, V> &m, T (&acc)[R][V])
{
    uint32_t size = m.G;
    for (uint32_t off = pow2_ceil(m.G) >> 1; off >= 1; off >>= 1) {
        T other[R][V];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int v = 0; v < V; ++v) other[i][v] = __shfl_down(acc[i][v], off * m.rpc, kWave);
        const bool take = m.g < off && m.g + off < size;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (take) acc[i][v] += other[i][v];
        size = off;
    }
}

// RowStream: stream the block-rows k = k_begin, k_begin + k_step, ... < k_end of one problem
// through this wavefront, in two calls so that the first matrix loads are already in flight while
// the caller still prepares the x operand (LDS staging, reductions, barriers):
//
//   prime(M, k_begin, k_end, k_step)   issue the first DEPTH units into the register ring.
//                                      Branch-free: exactly DEPTH*CH loads whatever the row count
//                                      (slots past the last unit re-read row 0, always in bounds).
//   run(xw, k_x0, N, on_row)           multiply; row k uses the LDS window xw + (k - k_x0)*n
//                                      (xw[c] = column c of row k_x0, zero-padded by the caller
//                                      where the vector ends).  on_row(k, acc) is called by every
//                                      lane after each row; lanes with g == 0 && active hold
//                                      y_k[rp*V + v] in acc[v].
//
// NCT > 0: the wave's rows are cut into units (StreamGeom) numbered q = 0, 1, ...; the ring keeps
// DEPTH-1 units of loads in flight behind the one being multiplied.  Loads return in order, the
// ring is always primed with the same number of loads and the steady-state loop refills every slot
// it consumes with no data-dependent control flow, so hipcc's counted vmcnt waits see exactly
// DEPTH-1 younger units (a conditionally primed ring makes its waitcnt pass assume the shortest
// queue and drain every iteration).  The ring slot is a compile-time index (loops unrolled by
// DEPTH); row and unit-in-row are runtime values that only enter address arithmetic.
// NCT == 0 (runtime n) has its own specialisation below with the same pipeline and runtime geometry.
template <typename T, int NCT, int V, bool NT = false> struct RowStream {
    using Gm = StreamGeom<T, NCT, V>;
    static constexpr int DEPTH = Gm::DEPTH;
    StreamUnit<T, NCT, V> ring[DEPTH];
    const T *M;
    uint32_t k_begin, k_end, k_step, total;

    __device__ __forceinline__ void issue(uint32_t q, int slot, const StreamCtx<T, NCT, V> &cx, uint32_t n) {
        const uint32_t ri = q / Gm::UPR, u = q - ri * Gm::UPR;
        const uint32_t k = q < total ? k_begin + ri * k_step : 0u;
        load_unit<T, NCT, V, NT>(M + (size_t)k * 3 * n * n, u, cx, ring[slot]);
    }

    __device__ __forceinline__ void prime(const T *__restrict__ M_, uint32_t k_begin_, uint32_t k_end_,
                                          uint32_t k_step_, const StreamCtx<T, NCT, V> &cx, uint32_t n) {
        M = M_;
        k_begin = k_begin_;
        k_end = k_end_;
        k_step = k_step_;
        const uint32_t nrows = k_end > k_begin ? (k_end - k_begin + k_step - 1) / k_step : 0;
        total = nrows * Gm::UPR;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            issue((uint32_t)j, j, cx, n);
            __builtin_amdgcn_sched_barrier(0);  // keep units in issue order: the waits count on it
        }
    }

    template <bool REFILL, typename RowFn>
    __device__ __forceinline__ void consume(uint32_t q, int slot, const T *xw, uint32_t k_x0, uint32_t N,
                                            const LaneMap<NCT, V> &m, const StreamCtx<T, NCT, V> &cx, bool refill,
                                            T (&acc)[V], RowFn &&on_row) {
        const uint32_t n = m.n;
        const uint32_t ri = q / Gm::UPR, u = q - ri * Gm::UPR;
        const uint32_t k = k_begin + ri * k_step;
        const uint32_t roff = (k - k_x0) * n;
        const T *xg = xw + cx.xcol + roff, *xl = xw + cx.xcol_last + roff;
        if (k == 0 || k == N - 1) {  // wave-uniform
            fma_unit<T, NCT, V, true>(ring[slot], u, xg, xl, cx, m.g, k == 0 ? n : 0u, k == N - 1 ? 2 * n : 3 * n, acc);
        } else {
            fma_unit<T, NCT, V, false>(ring[slot], u, xg, xl, cx, m.g, 0u, 3 * n, acc);
        }
        if (REFILL || refill) issue(q + DEPTH, slot, cx, n);
        if (Gm::UPR == 1 || u == Gm::UPR - 1) {
            fold_groups<T, NCT, V>(m, acc);
            on_row(k, acc);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = T(0);
        }
    }

    template <typename RowFn>
    __device__ __forceinline__ void run(const T *xw, uint32_t k_x0, uint32_t N, const LaneMap<NCT, V> &m,
                                        const StreamCtx<T, NCT, V> &cx, uint32_t /*lane*/, RowFn &&on_row) {
        T acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = T(0);
        uint32_t q0 = 0;
        // steady state: every consumed slot is refilled
        for (; q0 + 2 * DEPTH <= total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) consume<true>(q0 + j, j, xw, k_x0, N, m, cx, true, acc, on_row);
        }
        // drain: at most 2*DEPTH-1 units left
        for (; q0 < total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const uint32_t q = q0 + j;
                if (q < total) consume<false>(q, j, xw, k_x0, N, m, cx, q + DEPTH < total, acc, on_row);
            }
        }
    }
};

// ---- runtime block size (NCT == 0): the same register-ring pipeline with runtime geometry ----------
// Units are CH = 4 steps whatever n is (steps past the row's last one are masked: they re-read
// element 0 and contribute 0), so the number of loads per unit -- what the counted waits rely on --
// is still a compile-time constant; G, the step count and the units per row are runtime scalars.
template <typename T, int V> struct StreamCtx<T, 0, V> {
    uint32_t n, G, steps, upr, chunk;  // wave-uniform
    uint32_t g, off_lane;              // per lane
    bool active;
    static constexpr uint32_t CH = 4;
    __device__ __forceinline__ StreamCtx(const LaneMap<0, V> &m, uint32_t lane) {
        n = m.n; G = m.G; steps = m.steps; upr = (m.steps + CH - 1) / CH; chunk = m.G * m.n;
        g = m.g; active = m.active; off_lane = m.active ? lane * V : 0u;
    }
};

template <typename T, int V, bool NT> struct RowStream<T, 0, V, NT> {
    static constexpr uint32_t CH = StreamCtx<T, 0, V>::CH;
    static constexpr int DEPTH = (CH * V * sizeof(T) / 4) <= 8 ? 4 : 3;
    struct Unit { T a[CH][V]; };
    Unit ring[DEPTH];
    const T *M;
    uint32_t k_begin, k_end, k_step, total;
    uint32_t qi, ri_i, u_i;  // issue cursor: unit index, row index, unit-in-row

    __device__ __forceinline__ void issue_next(int slot, const StreamCtx<T, 0, V> &cx) {
        const uint32_t k = qi < total ? k_begin + ri_i * k_step : 0u;
        const T *Mk = M + (size_t)k * 3 * cx.n * cx.n;
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t s = u_i * CH + j;
            const uint32_t c = cx.g + cx.G * s;
            const bool ok = cx.active && s < cx.steps && c < 3 * cx.n;
            VecIO<T, V>::template load<NT>(Mk + (ok ? s * cx.chunk + cx.off_lane : 0u), ring[slot].a[j]);
        }
        ++qi;
        if (++u_i == cx.upr) { u_i = 0; ++ri_i; }
    }

    __device__ __forceinline__ void prime(const T *__restrict__ M_, uint32_t k_begin_, uint32_t k_end_,
                                          uint32_t k_step_, const StreamCtx<T, 0, V> &cx, uint32_t /*n*/) {
        M = M_; k_begin = k_begin_; k_end = k_end_; k_step = k_step_;
        const uint32_t nrows = k_end > k_begin ? (k_end - k_begin + k_step - 1) / k_step : 0;
        total = nrows * cx.upr;
        qi = 0; ri_i = 0; u_i = 0;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            issue_next(j, cx);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    template <typename RowFn>
    __device__ __forceinline__ void consume(int slot, uint32_t ri, uint32_t u, const T *xw, uint32_t k_x0, uint32_t N,
                                            const LaneMap<0, V> &m, const StreamCtx<T, 0, V> &cx, bool refill,
                                            T (&acc)[V], RowFn &&on_row) {
        const uint32_t n = cx.n;
        const uint32_t k = k_begin + ri * k_step;
        const T *xk = xw + (k - k_x0) * n;
        const uint32_t c_lo = k == 0 ? n : 0u, c_hi = k == N - 1 ? 2 * n : 3 * n;
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) {
            const uint32_t s = u * CH + j;
            const uint32_t c = cx.g + cx.G * s;
            const bool keep = cx.active && s < cx.steps && c >= c_lo && c < c_hi;
            const T xv = xk[c < 3 * n ? c : 3 * n - 1];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = fma_t(keep ? ring[slot].a[j][v] : T(0), xv, acc[v]);
        }
        if (refill) issue_next(slot, cx);
        if (u == cx.upr - 1) {
            fold_groups<T, 0, V>(m, acc);
            on_row(k, acc);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = T(0);
        }
    }

    template <typename RowFn>
    __device__ __forceinline__ void run(const T *xw, uint32_t k_x0, uint32_t N, const LaneMap<0, V> &m,
                                        const StreamCtx<T, 0, V> &cx, uint32_t /*lane*/, RowFn &&on_row) {
        T acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = T(0);
        uint32_t q0 = 0, ri = 0, u = 0;  // consume cursor
        for (; q0 + 2 * DEPTH <= total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                consume(j, ri, u, xw, k_x0, N, m, cx, true, acc, on_row);
                if (++u == cx.upr) { u = 0; ++ri; }
            }
        }
        for (; q0 < total; q0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (q0 + j < total) {
                    consume(j, ri, u, xw, k_x0, N, m, cx, q0 + j + DEPTH < total, acc, on_row);
                    if (++u == cx.upr) { u = 0; ++ri; }
                }
            }
        }
    }
};

// All-lanes sum of one value per lane; every lane returns the same total.
// DPP row shifts / row broadcasts in the VALU (no LDS crossbar round trips): partial sums run up
// each 16-lane row, rows 0,2 are broadcast into rows 1,3, row 1's total into the upper half, the
// wave total lands in lane 63 and is read back as a scalar.  Fixed, data-independent order.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    // lanes with no source (bound_ctrl off, old = 0) add 0
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}
template <int CTRL> __device__ __forceinline__ double dpp_add(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    const long long moved = ((long long)hi << 32) | (unsigned int)lo;
    return v + __builtin_bit_cast(double, moved);
}
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
    v = dpp_add<0x111>(v);  // row_shr:1
    v = dpp_add<0x112>(v);  // row_shr:2
    v = dpp_add<0x114>(v);  // row_shr:4
    v = dpp_add<0x118>(v);  // row_shr:8   -> lane 15 of each row holds the row total
    v = dpp_add<0x142>(v);  // row_bcast:15 -> rows 1 and 3 add the total of the row below
    v = dpp_add<0x143>(v);  // row_bcast:31 -> upper half adds the lower half's total
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    } else {
        const long long bits = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 63);
        const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
        return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned int)lo);
    }
}

// Round an element count up so the next LDS array stays 16-byte aligned (Guideline 17).
template <typename T> __host__ __device__ constexpr uint32_t align16(uint32_t elems)
{
    constexpr uint32_t q = 16 / sizeof(T);
    return (elems + q - 1) / q * q;
}

}  // namespace gbdpcg
