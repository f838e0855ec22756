// bt_device.hpp -- wavefront-level primitives for block-tridiagonal [L|D|R] matrices (gfx950).
//
// What replaces what (reference paths relative to /root/reference):
//   block_row_mv   <- loadbdVec + bdmv              (include/utils.cuh:9-85)
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
// n = 14 fp32 with 63/64 lanes live; 16 B for n = 36 fp64), each lane keeps ONE fixed row set
// and accumulates over columns in registers, and only a log2(G) shuffle tree is needed per
// block-row.  The x operand ([x_{k-1}; x_k; x_{k+1}], 3n values) is read from LDS as G
// broadcast addresses per step.  MFMA is not used: the contraction is a GEMV with
// arithmetic intensity ~0.5 flop/B, bounded by HBM.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gbdpcg {

constexpr uint32_t kWave = 64;

template <typename T, int V> struct VecOf;
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<double, 1> { using type = double; };
template <> struct VecOf<double, 2> { using type = double2; };

template <typename T, int V> struct VecIO;
template <typename T> struct VecIO<T, 1> {
    static __device__ __forceinline__ void load(const T *p, T (&a)[1]) { a[0] = *p; }
};
template <typename T> struct VecIO<T, 2> {
    static __device__ __forceinline__ void load(const T *p, T (&a)[2]) {
        using VT = typename VecOf<T, 2>::type;
        VT v = *reinterpret_cast<const VT *>(p);
        a[0] = v.x; a[1] = v.y;
    }
};
template <typename T> struct VecIO<T, 4> {
    static __device__ __forceinline__ void load(const T *p, T (&a)[4]) {
        using VT = typename VecOf<T, 4>::type;
        VT v = *reinterpret_cast<const VT *>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
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

// One block-row times [x_{k-1}; x_k; x_{k+1}].
//   Mk   : global, the 3n^2 elements of block-row k
//   xk   : LDS, xk[c] multiplies column c (c in [0,3n)); only c in [c_lo, c_hi) is touched
//          (c_lo = n for k = 0, c_hi = 2n for k = N-1: L_0 / R_{N-1} are never read,
//          include/pcg.cuh:105-106, include/utils.cuh:58-75)
// On return lanes with g == 0 (and active) hold y_k[rp*V + v] in acc[v]; other lanes hold junk.
template <typename T, int NCT, int V>
__device__ __forceinline__ void block_row_mv(const T *__restrict__ Mk, const T *xk,
                                             const LaneMap<NCT, V> &m, uint32_t lane,
                                             uint32_t c_lo, uint32_t c_hi, T (&acc)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = T(0);

    const uint32_t chunk = m.G * m.n;  // elements per step
    const T *src = Mk + lane * V;
    if (NCT) {
#pragma unroll
        for (uint32_t s = 0; s < m.steps; ++s) {
            const uint32_t c = m.g + m.G * s;
            if (m.active && c >= c_lo && c < c_hi) {
                T a[V];
                VecIO<T, V>::load(src + s * chunk, a);
                const T xv = xk[c];
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = __builtin_fma(a[v], xv, acc[v]);
            }
        }
    } else {
#pragma unroll 4
        for (uint32_t s = 0; s < m.steps; ++s) {
            const uint32_t c = m.g + m.G * s;
            if (m.active && c >= c_lo && c < c_hi) {
                T a[V];
                VecIO<T, V>::load(src + s * chunk, a);
                const T xv = xk[c];
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = __builtin_fma(a[v], xv, acc[v]);
            }
        }
    }

    // fold the G column-groups: halving tree over g with a lane stride of rpc
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

// All-lanes sum of one value per lane (butterfly; every lane returns the total).
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (uint32_t off = kWave / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Round an element count up so the next LDS array stays 16-byte aligned (Guideline 17).
template <typename T> __host__ __device__ constexpr uint32_t align16(uint32_t elems)
{
    constexpr uint32_t q = 16 / sizeof(T);
    return (elems + q - 1) / q * q;
}

}  // namespace gbdpcg
