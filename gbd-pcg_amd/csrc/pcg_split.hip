// pcg_split.hip -- PCG for problems too large for one workgroup's LDS (and for tiny batches):
// many workgroups per problem, TWO dependent launches per iteration, vectors in L2/HBM.
//
// Same algorithm as pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218).  The reference crosses
// the grid 4 times per iteration with cooperative-groups grid.sync(); on MI355X that barrier
// costs 26 us at 256 workgroups and even a hand-rolled XCD-hierarchical one 4-5 us, while a
// dependent kernel boundary costs ~1.5 us (MI355X_MICROARCH.md price list: barrier-cg,
// barrier-xcd, boundary).  So the iteration is cut at its two inner products and each half is
// one launch; the axpy-type updates are folded into the NEXT launch's prologue and recomputed
// redundantly for the one-knot halo, which removes the other two barriers:
//
//   k_init_r   : r = gamma - S lambda                                   (pcg.cuh:118-126)
//   k_precond  : [iter i>=0: alpha = eta/v; lambda += alpha p; r -= alpha ups]
//                r~ = Pinv r ; partial(r.r~)                            (pcg.cuh:130-149,172-193)
//   k_direction: eta' = sum partials; converged? ; beta = eta'/eta; p = r~ + beta p
//                ups = S p ; partial(p.ups)                             (pcg.cuh:195-206,156-165)
//   k_finish   : last convergence test, iters / max_iter_exit, copy r, p out   (pcg.cuh:212)
//
// Every workgroup sums the per-chunk partials itself in the same order, so all of them take the
// same convergence branch (the property pcg.cuh:147,167,191 rely on).  Vectors that a neighbour
// chunk still reads in the same launch are double-buffered by iteration parity.  A per-problem
// `done` word turns the launches after convergence into no-ops, which keeps the launch sequence
// static and hipGraph-capturable.
#include <chrono>
#include <cstdlib>
#include <thread>

#include "bt_device.hpp"
#include "internal.hpp"

namespace gbdpcg {

constexpr int kSplitWaves = 4;

// Per-problem workspace, in elements of T (then `batch` u32 done-words at the very end).
template <typename T> struct SplitWs {
    uint32_t len, npart;
    size_t r0, p0, ups, rt, pv, pe, per_problem;
    __host__ __device__ SplitWs(uint32_t n, uint32_t N) {
        len = n * N;
        npart = N;  // upper bound on chunks per problem
        const size_t L = align16<T>(len), Q = align16<T>(npart);
        r0 = 0;            // r[2][L]
        p0 = r0 + 2 * L;   // p[2][L]
        ups = p0 + 2 * L;  // ups[L]
        rt = ups + L;      // rt[L]
        pv = rt + L;       // pv[Q]
        pe = pv + Q;       // pe[2][Q]
        per_problem = pe + 2 * Q;
    }
};

template <typename T> size_t split_workspace_bytes(uint32_t n, uint32_t N, uint32_t batch)
{
    const SplitWs<T> w(n, N);
    size_t bytes = w.per_problem * sizeof(T) * batch;
    bytes = (bytes + 15) / 16 * 16;
    return bytes + (size_t)((batch + 3) / 4 * 4) * sizeof(uint32_t);
}

template <typename T> struct SplitArgs {
    PcgArgs<T> a;
    T *ws;
    uint32_t *done;
    uint32_t rpw, chunks;
};

// Sum `count` partials from global memory; every thread of every workgroup returns the same bits
// (fixed order: lane l adds elements l, l+64, ... ascending, then the wave butterfly), which is what
// keeps the exit test uniform across the grid.  `stage` is LDS scratch of >= count elements.
template <typename T>
__device__ __forceinline__ T sum_partials(const T *g, uint32_t count, T *stage, uint32_t tid, uint32_t nthreads)
{
    __syncthreads();
    for (uint32_t i = tid; i < count; i += nthreads) stage[i] = g[i];
    __syncthreads();
    T v = T(0);
    for (uint32_t i = tid & 63u; i < count; i += 64) v += stage[i];
    return wave_sum(v);
}

// Two partial arrays in one pass (one global round trip, one barrier pair): returns sum(gA) in
// *sa_out and sum(gB) in *sb_out.  `stage` holds 2*count elements.
template <typename T>
__device__ __forceinline__ void sum_partials2(const T *gA, const T *gB, uint32_t count, T *stage, uint32_t tid,
                                              uint32_t nthreads, T *sa_out, T *sb_out)
{
    __syncthreads();
    for (uint32_t i = tid; i < 2 * count; i += nthreads) stage[i] = i < count ? gA[i] : gB[i - count];
    __syncthreads();
    T va = T(0), vb = T(0);
    for (uint32_t i = tid & 63u; i < count; i += 64) {
        va += stage[i];
        vb += stage[count + i];
    }
    *sa_out = wave_sum(va);
    *sb_out = wave_sum(vb);
}

template <typename T, int WAVES>
__device__ __forceinline__ void store_partial(T part, T *dst, T *red, uint32_t lane, uint32_t wave)
{
    part = wave_sum(part);
    __syncthreads();
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        T tot = red[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) tot += red[w];
        *dst = tot;
    }
}

// LDS: window[(rpw+2)n] | stage[2*max(chunks, WAVES)] | red[WAVES] (16-byte aligned each)
template <typename T> __host__ __device__ inline size_t split_lds_elems(uint32_t n, uint32_t rpw, uint32_t chunks)
{
    return align16<T>((rpw + 2) * n) + align16<T>(2 * (chunks > (uint32_t)kSplitWaves ? chunks : kSplitWaves)) +
           align16<T>(kSplitWaves);
}

enum SplitPhase { PH_INIT_R = 0, PH_PRECOND = 1, PH_DIRECTION = 2 };

// One launch = one phase over all (problem, chunk) workgroups.
//   PH_INIT_R   : window = lambda                      ; out = gamma - S*window -> r[1]
//   PH_PRECOND  : window = r_old - alpha*ups (or r[1] when iter == -1, the prologue)
//                 own rows -> r[iter&1], lambda += alpha p ; rt = Pinv*window ; pe[iter&1][chunk]
//   PH_DIRECTION: window = rt + beta*p_old (or rt when iter == 0) ; own rows -> p[iter&1]
//                 ups = S*window ; pv[chunk]
template <typename T, int NCT, int V, int PHASE>
__global__ __launch_bounds__(kSplitWaves * 64) void pcg_split_kernel(SplitArgs<T> sa, int iter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WAVES = kSplitWaves;
    constexpr uint32_t THREADS = WAVES * 64;
    const PcgArgs<T> &a = sa.a;
    const uint32_t n = NCT ? (uint32_t)NCT : a.n;
    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps row bases in SGPRs
    const uint32_t prob = blockIdx.x / sa.chunks;
    const uint32_t chunk = blockIdx.x - prob * sa.chunks;
    const uint32_t k0 = chunk * sa.rpw, k1 = min(N, k0 + sa.rpw);

    // progress beacon for the blocking host loop (launch_split_v): the direction launch of iteration `iter` has started
    if (PHASE == PH_DIRECTION && a.host_done && blockIdx.x == 0 && tid == 0)
        __hip_atomic_store(a.host_done + 1, (uint32_t)iter + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (PHASE != PH_INIT_R && sa.done[prob]) return;

    // matrix loads first: they depend on no vector, so they fly while partials are summed and the
    // x window is built
    const LaneMap<NCT, V> m(n, lane);
    const StreamCtx<T, NCT, V> cx(m, lane);
    RowStream<T, NCT, V> rs;
    {
        const size_t ms = (size_t)3 * n * n * N;
        const T *M0 = (PHASE == PH_PRECOND) ? (a.Pinv ? a.Pinv + prob * ms : a.S + prob * ms) : a.S + prob * ms;
        rs.prime(M0, k0 + wave, k1, WAVES, cx, n);
    }

    T *win = reinterpret_cast<T *>(smem_raw);
    T *stage = win + align16<T>((sa.rpw + 2) * n);
    T *red = stage + align16<T>(2 * (sa.chunks > (uint32_t)WAVES ? sa.chunks : WAVES));

    const SplitWs<T> w(n, N);
    T *ws = sa.ws + (size_t)prob * w.per_problem;
    const size_t L = align16<T>(len), Q = align16<T>(w.npart);
    T *rbuf[2] = {ws + w.r0, ws + w.r0 + L};
    T *pbuf[2] = {ws + w.p0, ws + w.p0 + L};
    T *ups = ws + w.ups, *rt = ws + w.rt, *pv = ws + w.pv;
    T *pe[2] = {ws + w.pe, ws + w.pe + Q};
    const size_t mstride = (size_t)3 * n * n * N;
    const T *S = a.S + prob * mstride;
    const T *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;
    T *lambda = a.lambda + (size_t)prob * len;

    const uint32_t cnt = (k1 - k0 + 2) * n;
    const int64_t g0 = (int64_t)k0 * n - n;
    const uint32_t own_lo = n, own_hi = n + (k1 - k0) * n;  // window indices of this chunk's own rows
    const int par = iter & 1;

    const T *M = nullptr;
    T *out = nullptr;
    if (PHASE == PH_INIT_R) {
        for (uint32_t i = tid; i < cnt; i += THREADS) {
            const int64_t gi = g0 + i;
            win[i] = (gi >= 0 && gi < (int64_t)len) ? lambda[gi] : T(0);
        }
        M = S;
    } else if (PHASE == PH_PRECOND) {
        if (iter < 0) {  // prologue: r already complete in r[1]
            for (uint32_t i = tid; i < cnt; i += THREADS) {
                const int64_t gi = g0 + i;
                win[i] = (gi >= 0 && gi < (int64_t)len) ? rbuf[1][gi] : T(0);
            }
        } else {
            // eta = r.r~ of the previous half-step lives in pe[par^1]; v = p.ups in pv.
            // The first window element of every thread is fetched BEFORE the partial sums: the two
            // global round trips overlap instead of following each other.
            const T *r_old = rbuf[par ^ 1];
            const T *p_cur = pbuf[par];
            const int64_t gi0 = g0 + tid;
            const bool in0 = tid < cnt && gi0 >= 0 && gi0 < (int64_t)len;
            const bool own0 = in0 && tid >= own_lo && tid < own_hi;
            const T r0 = in0 ? r_old[gi0] : T(0), u0 = in0 ? ups[gi0] : T(0);
            const T p0 = own0 ? p_cur[gi0] : T(0), l0 = own0 ? lambda[gi0] : T(0);
            T eta, v;
            sum_partials2(pe[par ^ 1], pv, sa.chunks, stage, tid, THREADS, &eta, &v);
            const T alpha = eta / v;
            if (tid < cnt) {
                const T rv = in0 ? fma_t(-alpha, u0, r0) : T(0);
                if (own0) {
                    rbuf[par][gi0] = rv;
                    lambda[gi0] = fma_t(alpha, p0, l0);
                }
                win[tid] = rv;
            }
            for (uint32_t i = tid + THREADS; i < cnt; i += THREADS) {
                const int64_t gi = g0 + i;
                T rv = T(0);
                if (gi >= 0 && gi < (int64_t)len) {
                    rv = fma_t(-alpha, ups[gi], r_old[gi]);
                    if (i >= own_lo && i < own_hi) {
                        rbuf[par][gi] = rv;
                        lambda[gi] = fma_t(alpha, p_cur[gi], lambda[gi]);
                    }
                }
                win[i] = rv;
            }
        }
        M = P;
        out = rt;
    } else {  // PH_DIRECTION
        if (iter == 0) {
            for (uint32_t i = tid; i < cnt; i += THREADS) {
                const int64_t gi = g0 + i;
                T pn = T(0);
                if (gi >= 0 && gi < (int64_t)len) {
                    pn = rt[gi];
                    if (i >= own_lo && i < own_hi) pbuf[0][gi] = pn;
                }
                win[i] = pn;
            }
        } else {
            const T *p_old = pbuf[par ^ 1];
            const int64_t gi0 = g0 + tid;
            const bool in0 = tid < cnt && gi0 >= 0 && gi0 < (int64_t)len;
            const T q0 = in0 ? p_old[gi0] : T(0), t0 = in0 ? rt[gi0] : T(0);  // fetched before the sums
            T eta_new, eta;
            sum_partials2(pe[par ^ 1], pe[par], sa.chunks, stage, tid, THREADS, &eta_new, &eta);
            if (fabs(eta_new) < a.tol) {  // iteration iter-1 converged (pcg.cuh:195)
                if (chunk == 0 && tid == 0) {
                    sa.done[prob] = 1;
                    a.iters[prob] = (uint32_t)iter;
                    if (a.max_iter_exit) a.max_iter_exit[prob] = 0;
                    if (a.host_done) __hip_atomic_fetch_add(a.host_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return;
            }
            const T beta = eta_new / eta;
            if (tid < cnt) {
                const T pn = in0 ? fma_t(beta, q0, t0) : T(0);
                if (in0 && tid >= own_lo && tid < own_hi) pbuf[par][gi0] = pn;
                win[tid] = pn;
            }
            for (uint32_t i = tid + THREADS; i < cnt; i += THREADS) {
                const int64_t gi = g0 + i;
                T pn = T(0);
                if (gi >= 0 && gi < (int64_t)len) {
                    pn = fma_t(beta, p_old[gi], rt[gi]);
                    if (i >= own_lo && i < own_hi) pbuf[par][gi] = pn;
                }
                win[i] = pn;
            }
        }
        M = S;
        out = ups;
    }
    __syncthreads();

    T part = T(0);
    if (PHASE == PH_PRECOND && M == nullptr) {
        // identity preconditioner: r~ = r
        for (uint32_t i = own_lo + tid; i < own_hi; i += THREADS) {
            const T rv = win[i];
            out[g0 + i] = rv;
            part = fma_t(rv, rv, part);
        }
    } else {
        rs.run(win, k0, N, m, cx, lane, [&](uint32_t k, const T(&acc)[V]) __attribute__((always_inline)) {
            if (m.active && m.g == 0) {
                const uint32_t row = k * n + m.rp * V;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (PHASE == PH_INIT_R) {
                        rbuf[1][row + v] = a.gamma[(size_t)prob * len + row + v] - acc[v];
                    } else {
                        out[row + v] = acc[v];
                        part = fma_t(acc[v], win[n + (row - k0 * n) + v], part);
                    }
                }
            }
        });
    }
    if (PHASE == PH_PRECOND) store_partial<T, WAVES>(part, &pe[par][chunk], red, lane, wave);
    if (PHASE == PH_DIRECTION) store_partial<T, WAVES>(part, &pv[chunk], red, lane, wave);
}

// After the last iteration: final convergence test (pcg.cuh:195 for iteration max_iter-1),
// iters / max_iter_exit (pcg.cuh:212), and r, p copied to the caller's buffers.
// grid = batch * fchunks workgroups; each redoes the (cheap) test and copies its 1024-element slice.
template <typename T>
__global__ __launch_bounds__(256) void pcg_split_finish(SplitArgs<T> sa, uint32_t fchunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *stage = reinterpret_cast<T *>(smem_raw);
    const PcgArgs<T> &a = sa.a;
    const uint32_t n = a.n, N = a.N, len = n * N;
    const uint32_t prob = blockIdx.x / fchunks, fc = blockIdx.x - prob * fchunks, tid = threadIdx.x;
    const SplitWs<T> w(n, N);
    T *ws = sa.ws + (size_t)prob * w.per_problem;
    const size_t L = align16<T>(len), Q = align16<T>(w.npart);

    uint32_t iters;
    bool update_p = false;  // the last iteration did not break, so it still ran p = r~ + beta p (pcg.cuh:203-206)
    T beta = T(0);
    if (sa.done[prob]) {
        iters = a.iters[prob];
    } else {
        iters = a.max_iter;
        bool exit_flag = true;
        if (a.max_iter > 0) {
            T eta_new, eta;
            sum_partials2(ws + w.pe + ((a.max_iter - 1) & 1) * Q, ws + w.pe + (a.max_iter & 1) * Q, sa.chunks, stage, tid,
                          256u, &eta_new, &eta);
            exit_flag = !(fabs(eta_new) < a.tol);
            if (exit_flag) {
                beta = eta_new / eta;
                update_p = true;
            }
        }
        if (fc == 0 && tid == 0) {
            a.iters[prob] = iters;
            if (a.max_iter_exit) a.max_iter_exit[prob] = exit_flag ? 1 : 0;
        }
    }
    // state left behind: r after `iters` residual updates; p after the direction updates that ran
    const T *r_fin = ws + w.r0 + (iters == 0 ? 1 : ((iters - 1) & 1)) * L;
    const T *p_last = iters == 0 ? ws + w.rt : ws + w.p0 + ((iters - 1) & 1) * L;
    const T *rt = ws + w.rt;
    const uint32_t lo = fc * 1024u, hi = min(len, lo + 1024u);
    for (uint32_t i = lo + tid; i < hi; i += 256) {
        if (a.r) a.r[(size_t)prob * len + i] = r_fin[i];
        if (a.p) a.p[(size_t)prob * len + i] = update_p ? fma_t(beta, p_last[i], rt[i]) : p_last[i];
    }
}

template <typename T, int NCT, int V>
static hipError_t launch_split_v(const DeviceInfo &dev, const PcgArgs<T> &a, void *workspace, hipStream_t s,
                                 const volatile uint32_t *poll)
{
    constexpr int WAVES = kSplitWaves;
    SplitArgs<T> sa;
    sa.a = a;
    sa.ws = reinterpret_cast<T *>(workspace);
    const SplitWs<T> w(a.n, a.N);
    size_t vec_bytes = (w.per_problem * sizeof(T) * a.batch + 15) / 16 * 16;
    sa.done = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(workspace) + vec_bytes);

    // ~4 workgroups per CU over all problems, whole block-rows per wave, and never fewer rows than
    // waves in a workgroup (a single large problem then uses fewer, fully busy workgroups and has
    // fewer partials to sum)
    const uint64_t total_rows = (uint64_t)a.N * a.batch;
    const uint64_t target = (uint64_t)dev.num_cus * 4;
    uint32_t rpw = (uint32_t)((total_rows + target - 1) / target);
    if (rpw < (uint32_t)WAVES) rpw = WAVES;
    rpw = (rpw + WAVES - 1) / WAVES * WAVES;
    if (rpw > a.N) rpw = a.N;
    if (rpw == 0) rpw = 1;
    while (split_lds_elems<T>(a.n, rpw, (a.N + rpw - 1) / rpw) * sizeof(T) > dev.lds_per_wg_max && rpw > 1) rpw /= 2;
    sa.rpw = rpw;
    sa.chunks = (a.N + rpw - 1) / rpw;
    const size_t lds = split_lds_elems<T>(a.n, sa.rpw, sa.chunks) * sizeof(T);
    if (lds > dev.lds_per_wg_max) return hipErrorInvalidValue;

    auto k_init = pcg_split_kernel<T, NCT, V, PH_INIT_R>;
    auto k_pre = pcg_split_kernel<T, NCT, V, PH_PRECOND>;
    auto k_dir = pcg_split_kernel<T, NCT, V, PH_DIRECTION>;
    if (lds > 64 * 1024) {
        const void *ks[] = {(const void *)k_init, (const void *)k_pre, (const void *)k_dir};
        for (const void *k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
    }
    hipError_t e = launch_fill_words(sa.done, 0u, (size_t)((a.batch + 3) / 4 * 4), s);
    if (e != hipSuccess) return e;

    const dim3 grid(sa.chunks * a.batch), block(WAVES * 64);
    hipLaunchKernelGGL(k_init, grid, block, lds, s, sa, 0);
    hipLaunchKernelGGL(k_pre, grid, block, lds, s, sa, -1);
    for (uint32_t it = 0; it < a.max_iter; ++it) {
        // Eager blocking solves are bound by the host's launch rate (3.5 us per launch against 5 us per
        // iteration on the device): stop enqueueing once every problem has reported convergence.  The
        // launches skipped would have been no-ops (done[prob] is set), so the result is the same.
        if (poll) {
            // ... and never run more than kAhead iterations ahead of the device, or everything is enqueued
            // long before the first convergence report arrives (poll[1] = last iteration the device started)
            // The wait is bounded in TIME (a launch pair takes ~12 us; 2 ms without progress means the stream is stuck
            // behind other work, and then running ahead is harmless) and gives the core away between looks.
            constexpr uint32_t kAhead = 3;
            if (it >= kAhead) {
                const auto t0 = std::chrono::steady_clock::now();
                for (uint32_t looks = 0; poll[1] + kAhead < it + 1 && poll[0] < a.batch; ++looks) {
                    if (looks < 64) continue;   // the device is usually one launch behind: a few hundred ns
                    std::this_thread::yield();
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                }
            }
            if (poll[0] >= a.batch) break;
        }
        hipLaunchKernelGGL(k_dir, grid, block, lds, s, sa, (int)it);
        hipLaunchKernelGGL(k_pre, grid, block, lds, s, sa, (int)it);
    }
    const size_t lds_fin = align16<T>(2 * sa.chunks) * sizeof(T);
    const uint32_t fchunks = ((uint32_t)a.n * a.N + 1023u) / 1024u;
    hipLaunchKernelGGL(pcg_split_finish<T>, dim3(a.batch * fchunks), dim3(256), lds_fin, s, sa, fchunks);
    return hipGetLastError();
}

template <typename T, int NCT>
static hipError_t launch_split_n(const DeviceInfo &dev, const PcgArgs<T> &a, void *ws, int V, hipStream_t s,
                                 const volatile uint32_t *poll)
{
    if (V == 1) return launch_split_v<T, NCT, 1>(dev, a, ws, s, poll);
    if constexpr (NCT == 0 || NCT % 2 == 0) {
        if (V == 2) return launch_split_v<T, NCT, 2>(dev, a, ws, s, poll);
    }
    if constexpr (sizeof(T) == 4 && (NCT == 0 || NCT % 4 == 0)) {
        if (V == 4) return launch_split_v<T, NCT, 4>(dev, a, ws, s, poll);
    }
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_pcg_split(const DeviceInfo &dev, const PcgArgs<T> &a, void *workspace, hipStream_t s,
                            const volatile uint32_t *poll)
{
    const void *ptrs[] = {a.S, a.Pinv};
    const int V = choose_vec<T>(a.n, ptrs, 2);
    if (V == 0 || workspace == nullptr) return hipErrorInvalidValue;
    static const bool generic_only = getenv("GBDPCG_FORCE_GENERIC") != nullptr;  // tuning runs only
    if (!generic_only) {
#define GBDPCG_CASE(NN) \
    if (a.n == NN && V == best_v<T, NN>()) return launch_split_v<T, NN, best_v<T, NN>()>(dev, a, workspace, s, poll);
        GBDPCG_SPECIALIZED_N(GBDPCG_CASE)
#undef GBDPCG_CASE
    }
    return launch_split_n<T, 0>(dev, a, workspace, V, s, poll);
}

template size_t split_workspace_bytes<float>(uint32_t, uint32_t, uint32_t);
template size_t split_workspace_bytes<double>(uint32_t, uint32_t, uint32_t);
template hipError_t launch_pcg_split<float>(const DeviceInfo &, const PcgArgs<float> &, void *, hipStream_t,
                                            const volatile uint32_t *);
template hipError_t launch_pcg_split<double>(const DeviceInfo &, const PcgArgs<double> &, void *, hipStream_t,
                                             const volatile uint32_t *);

}  // namespace gbdpcg
