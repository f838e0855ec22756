// pcg_cluster.hip -- PCG with both GENERAL-storage matrices register-resident, a problem spread over a CLUSTER of 2-4 CUs.
//
// Replaces pcg<T,n,N> (/root/reference/include/pcg.cuh:54-218) for the BASELINE batch shape (n = 14, fp32, N = 128) when
// the storage is not bit-symmetric (pcg_resident_sym.hip does not apply) and for 72 < N <= 288 in general.  The reference
// keeps all three blocks [L|D|R] of a knot of S and Pinv next to the ALUs for the whole solve (pcg.cuh:104-110); at this
// shape that is 602 KB per problem, more than one CU holds (512 KB of registers + 160 KB of LDS), so the problem is cut
// into H = ceil(N / 72) runs of consecutive knots and each run is given to one 8-wave workgroup with the lane map of
// pcg_resident.hip (bt_dense.hpp: 2 x 84 VGPRs of matrix data per lane, no cross-lane fold).  The matrices are read from
// HBM once per solve; an iteration touches LDS, registers and the hand-off words below.
//
// What crosses between the workgroups of a cluster, twice per iteration (the two inner products of PCG depend on each
// other) and in ONE go each time: the 8 wave partials of the inner product and the boundary knot of the product vector
// that the neighbour needs; the neighbour then updates its halo copy of r / p with the same fma the owner uses (same bits).
// Every workgroup sums all 8H partials in the same order -> the exit test of pcg.cuh:195 is uniform over the cluster --
// what pcg.cuh:147,167,191 ensure with their redundant per-block sums.  Hand-off words are data-tagged 8-byte granules
// {value, epoch} written with sc1 (write-through) stores and polled with sc1 loads: the data is the flag, no fence
// (MI355X_MICROARCH.md, Valid forms, R2; the same protocol as pcg_persist.hip).  One such hand-off between two CUs costs
// 0.44-0.56 us (tools/hop_probe.hip), against 1.4-2 us for the chip-wide all-gather of the persistent path.
//
// Nothing is initialised or cleared by a launch (a solve stays ONE kernel node in a hipGraph) and a slot is only ever
// written by the workgroup that owns it.  Epochs start at 1 in every launch, and a tag is {launch number, epoch}: the
// launch number comes from a word of the workspace that the workgroup finishing LAST (an agent-scope counter tells it)
// increments, so every workgroup of a launch reads the same one, and whatever an earlier launch left in a slot -- or in
// some L2 -- carries another launch number and is never taken for a publication of this one.  (Two earlier forms: the
// last finisher zeroing all slots -- it wrote lines from one XCD that another XCD had written, which rules out the
// plain stores below; then every owner zeroing its own slots behind a closing hand-off -- on one box a launch that
// followed a launch with another block-to-XCD mapping accepted stale granules, presumably out of an L2 line that had
// outlived the kernel boundary; with the launch number in the tag such a line is harmless.)
// Before its first problem a cluster says HELLO (epoch 1: every member publishes the id of the XCD it runs on).
// When HELLO shows every member of a cluster on the SAME XCD (members sit 8 blocks apart for that, under round-robin
// dispatch) the cluster publishes with PLAIN stores: the line stays in the XCD's L2, where the partner's sc1 loads find
// it -- 0.25 us per hand-off instead of 0.44 (tools/hop_probe.hip; across XCDs a plain store is never seen, hence the
// check).  Otherwise, and always across XCDs, stores are sc1 (write-through).  Nothing depends on placement.
// A cluster's workgroups must be resident together: the grid never exceeds one workgroup per CU, members of a cluster have
// neighbouring block indices (in-order dispatch then splits at most one cluster at a time, and that one only until any
// workgroup exits), and every spin is bounded: a cluster that cannot complete a hand-off gives its remaining problems up
// -- nothing of them is written -- and the member that LEAVES LAST (a per-cluster agent-scope counter tells it; the
// others have left by then) solves them alone, streaming, inside this same launch (pcg_stream.hpp, stream_rescue): no
// caller ever sees an unsolved problem, and a healthy launch pays one atomic per workgroup for it.
// A tag is {launch number mod 4095, + 1 : 12 bits | epoch : 20 bits} -- a fixed split, so launches with different
// max_iter cannot produce each other's tags -- and every workgroup of a launch that owns a problem clears its own slot
// (both parities) before HELLO, so a granule outlives at most the launches that own nothing: a stale one can only carry
// the tag a launch polls for if its launch number is 4095 k launches old AND no launch in between owned a problem.
#include <cstdlib>

#include "bt_dense.hpp"
#include "pcg_stream.hpp"

namespace gbdpcg {

namespace {

typedef unsigned long long u64;
typedef unsigned int cl_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int cl_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kClSc1 = 16;   // cache-policy bits of the raw buffer builtins on gfx950: bit 4 = sc1

// Workspace: [256-byte block: stamps of the diagnostic build, words 30 / 31 = finish counter / launch number |
// per cluster two words: members that have left, and 2^32 - 1 - (first problem a member gave up on) maximised over the
// members (0: none) | slots[2 parities][grid blocks]].  One slot = three 128-byte lines:
//   +0    the 8 wave partials (8 granules)
//   +128  the first own knot of the product vector, for the left neighbour  (n granules, two per 16-byte store)
//   +256  the last own knot, for the right neighbour
constexpr uint32_t kClLeftOff = 256, kClCtrlBytes = 256 + 256 * 16, kClSlotBytes = 384, kClFirstOff = 128, kClLastOff = 256;
constexpr uint32_t kClMaxH = 4;
constexpr uint32_t kClEpochBits = 20, kClLaunchMod = 4095;   // tag = ((launch mod 4095) + 1) << 20 | epoch

__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }

// Lane roles of the polling wave in one gather: lanes [0, 4H) fetch two wave partials each (member L / 4, waves
// 2 (L % 4), +1), lanes [32, 39) the left neighbour's last knot, lanes [40, 47) the right neighbour's first knot.
constexpr uint32_t kClLeftLane = 32, kClRightLane = 40;

#ifndef GBDPCG_CL_PTAIL
#define GBDPCG_CL_PTAIL 0   // columns of a lane's Pinv block-row kept in LDS instead of registers (0, 2, 4, ...): not needed since the kernel compiles without scratch; kept for A/B builds
#endif
template <int NCT> struct ClusterTail { static constexpr int COLS = NCT == 16 ? 16 : GBDPCG_CL_PTAIL; };
#ifndef GBDPCG_CL_CHAINS
#define GBDPCG_CL_CHAINS 3   // accumulator chains of a block-row product (bt_dense.hpp, dense_mv); 1 for A/B builds
#endif

}  // namespace

template <int NCT, int V, bool STAGED>
__global__ __launch_bounds__(512) void pcg_cluster_kernel(PcgArgs<float> a, unsigned char *ws, uint32_t H, uint32_t C,
                                                          uint32_t clusters, uint32_t spin_limit, uint32_t drop_block, bool no_plain)
{
#ifndef GBDPCG_TEST_HOOKS
    drop_block = 0xffffffffu;   // the hook that silences one workgroup exists in variants/libgbdpcg_hooks.so only
#endif
    constexpr uint32_t epoch_bits = kClEpochBits;
    using Dg = DenseGeom<float, NCT, V>;
    // lane roles below: n / 2 lanes x 2 rows per knot (a knot's boundary values are one 128-byte line of 16-byte granule pairs:
    // n / 2 <= 8), 8 waves (lanes 4H..: partials of the 8 H waves, two per lane)
    static_assert(V == 2 && NCT % 2 == 0 && NCT <= 16 && Dg::WAVES == 8, "lane roles below are written for n / 2 lanes x 2 rows per knot");
    constexpr uint32_t n = NCT, THREADS = Dg::WAVES * 64, WINF = align16<float>((Dg::MAX_KNOTS + 2) * n);
    __shared__ __attribute__((aligned(16))) float xa[WINF];   // window of p (lambda in the prologue): halo knot, own knots, halo knot
    __shared__ __attribute__((aligned(16))) float xb[WINF];   // window of r
    __shared__ float bc[4];       // [0] alpha / eta' of the phase just gathered, [1] beta
    __shared__ uint32_t bci[4];   // [0] 0 go on, 1 converged, 2 hand-off timed out; [2], [3] the rescue's bookkeeping
    __shared__ float rescue_red[2 * Dg::WAVES];
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];   // STAGED: dense_stage_lds_bytes (bt_dense.hpp)
    // The last PTAIL columns of this lane's block-row of Pinv (two rows each) live in LDS instead of registers (dense_mv, TAIL):
    // at n = 16 the whole R block of Pinv -- 2 x 96 matrix registers per lane do not fit next to the working set, 96 + 64 do.
    // They take the place of the staging buffers once every wave's tiles are in (cluster_lds_bytes).
    constexpr int PTAIL = ClusterTail<NCT>::COLS;
    float2 *ptail = reinterpret_cast<float2 *>(stage_raw);

    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t grid = gridDim.x, blk = blockIdx.x;
    // cluster c, member h.  Members sit 8 blocks apart where the cluster count allows it: blocks b and b + 8 share an XCD
    // under round-robin dispatch (a hand-off inside one L2 is ~20 % shorter).  Speed only: nothing depends on placement.
    const bool spread = clusters % 8 == 0;
    const uint32_t c = spread ? (blk / (8 * H)) * 8 + blk % 8 : blk / H;
    const uint32_t h = spread ? (blk % (8 * H)) / 8 : blk % H;
    const uint32_t bstride = spread ? 8u : 1u;   // block index distance of two neighbouring members
    const uint32_t blk0 = blk - h * bstride;     // member 0 of this cluster
    const uint32_t k_lo = h * C, cnt = k_lo < N ? (N - k_lo < C ? N - k_lo : C) : 0u;   // host: cnt >= 1 for every member
    const bool has_left = h > 0, has_right = h + 1 < H;

    const DenseCtx<float, NCT, V> dc(wave, lane, cnt, k_lo);
    // first of this lane's rows, counted from knot k_lo: (wave * 9 + lane / 7) * 14 + (lane % 7) * 2 = wave * 126 + 2 * lane
    const uint32_t row0 = dc.live ? wave * (Dg::BPW * n) + 2 * lane : 0u;
    // the same index inside a window, from an opaque copy of the lane number (see GBDPCG_CL_HANDOFF: not worth a register)
    auto own_idx = [&]() {
        uint32_t lo = lane;
        asm volatile("" : "+v"(lo));
        return n + wave * (Dg::BPW * n) + 2 * lo;
    };
    const uint32_t wl = (cnt - 1) / Dg::BPW, lb = (cnt - 1) - wl * Dg::BPW;   // the last knot: wave wl, lanes [7 lb, 7 lb + 7)
    const uint32_t POLL = wl == 7 ? 6u : 7u;                          // the polling wave: never wave 0, never wave wl
    const size_t mstride = (size_t)3 * n * n * N;

    unsigned char *slots = ws + kClCtrlBytes;
    const __amdgpu_buffer_rsrc_t region =
        __builtin_amdgcn_make_buffer_rsrc(slots, 0, (int)(2u * grid * kClSlotBytes), 0x00020000);
    const uint32_t my_slot = blk * kClSlotBytes, par_stride = grid * kClSlotBytes;

    // LDS is workgroup-private and the hand-off stores must not be waited for: a barrier that drains only the LDS counter
#ifdef GBDPCG_CL_SYNCTHREADS   // diagnostic variant: the full barrier (also waits for the hand-off stores)
    auto wg_barrier = [] { __syncthreads(); };
#else
    auto wg_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif

    // One hand-off.  Every wave publishes its share PART of the inner product; the lanes that own the first / last knot
    // publish the two values V0, V1 of their rows.  The polling wave then gathers the 8H partials and the neighbours'
    // boundary knots: TOTAL (wave-uniform, the same bits in every member) and, in its halo lanes, G0 / G1; HIDX = the
    // entry of window HWIN a halo lane rewrites afterwards (0xffffffff: not a halo lane), H0 / H1 its current content.
    // What a lane does follows from its lane number alone; it is recomputed here from an opaque copy of it, every time:
    // kept in registers across the products (168 of 256 VGPRs hold matrix data) these few values were spilled, and each
    // reload from scratch sat, with its s_waitcnt vmcnt(0), in front of the very stores the neighbour is waiting for.
#define GBDPCG_CL_HANDOFF(EPOCH, PART, V0, V1, HWIN, TOTAL, G0, G1, H0, H1, HIDX, OK)                               \
    {                                                                                                                \
        const uint32_t tag = nonce | (EPOCH), par_off = ((EPOCH) & 1u) * par_stride;                                 \
        uint32_t lo = lane;                                                                                          \
        asm volatile("" : "+v"(lo));                                                                                 \
        if (blk != drop_block) {                                                                                     \
            const cl_u32x2 x = {fbits(PART), tag};                                                                   \
            const cl_u32x4 bx = {fbits(V0), tag, fbits(V1), tag};                                                    \
            const int o_part = (int)(par_off + my_slot + wave * 8), o_first = (int)(par_off + my_slot + kClFirstOff + lo * 16), \
                      o_last = (int)(par_off + my_slot + kClLastOff + (lo - lb * (n / 2)) * 16);                     \
            const bool p_first = wave == 0 && has_left && lo < n / 2, p_last = wave == wl && has_right && lo - lb * (n / 2) < n / 2; \
            if (same_xcd) {   /* plain: the line stays in this XCD's L2, where every member of the cluster polls */   \
                if (lo == 0) __builtin_amdgcn_raw_buffer_store_b64(x, region, o_part, 0, 0);                         \
                if (p_first) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_first, 0, 0);                      \
                if (p_last) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_last, 0, 0);                        \
            } else {          /* sc1: write-through, seen from any XCD */                                            \
                if (lo == 0) __builtin_amdgcn_raw_buffer_store_b64(x, region, o_part, 0, kClSc1);                    \
                if (p_first) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_first, 0, kClSc1);                 \
                if (p_last) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_last, 0, kClSc1);                   \
            }                                                                                                        \
        }                                                                                                            \
        if (wave == POLL) {                                                                    \
            uint32_t poll_off = 0;                                                                                   \
            bool have = true;                                                                                        \
            HIDX = 0xffffffffu;                                                                                      \
            if (lo < 4 * H) {                                                                                        \
                poll_off = (blk0 + (lo >> 2) * bstride) * kClSlotBytes + (lo & 3u) * 16;                             \
                have = false;                                                                                        \
            } else if (lo - kClLeftLane < n / 2 && has_left) {                                                       \
                poll_off = (blk - bstride) * kClSlotBytes + kClLastOff + (lo - kClLeftLane) * 16;                    \
                HIDX = (lo - kClLeftLane) * 2;                                                                       \
                have = false;                                                                                        \
            } else if (lo - kClRightLane < n / 2 && has_right) {                                                     \
                poll_off = (blk + bstride) * kClSlotBytes + kClFirstOff + (lo - kClRightLane) * 16;                  \
                HIDX = (cnt + 1) * n + (lo - kClRightLane) * 2;                                                      \
                have = false;                                                                                        \
            }                                                                                                        \
            /* the halo entries this lane is going to update: read now, under the wait */                           \
            if (HIDX != 0xffffffffu) {                                                                               \
                H0 = (HWIN)[HIDX];                                                                                   \
                H1 = (HWIN)[HIDX + 1];                                                                               \
            }                                                                                                        \
            cl_u32x4 got = {0u, 0u, 0u, 0u};                                                                         \
            OK = true;                                                                                               \
            for (uint32_t spins = 0;; ++spins) {                                                                     \
                asm volatile("" ::: "memory");   /* the poll is re-issued on every pass */                           \
                if (!have) {   /* only the lanes that still miss their piece load again */                           \
                    got = __builtin_amdgcn_raw_buffer_load_b128(region, (int)(par_off + poll_off), 0, kClSc1);       \
                    have = got.y == tag && got.w == tag;                                                             \
                }                                                                                                    \
                if (__all(have)) break;                                                                              \
                if (spins >= spin_limit) {                                                                           \
                    OK = false;                                                                                      \
                    break;                                                                                           \
                }                                                                                                    \
                __builtin_amdgcn_s_sleep(1);                                                                         \
            }                                                                                                        \
            /* by value: __builtin_bit_cast applied to the element expression got.z itself reads element 0 (hipcc,   \
               ROCm 7.2) */                                                                                          \
            const uint32_t gx = got.x, gz = got.z;                                                                   \
            G0 = __builtin_bit_cast(float, gx);                                                                      \
            G1 = __builtin_bit_cast(float, gz);                                                                      \
            TOTAL = wave_sum(lo < 4 * H ? G0 + G1 : 0.f);                                                            \
        }                                                                                                            \
    }

    // Diagnostic build only (-DGBDPCG_CL_STAMPS, tools/cluster_stamps.py): cycle stamps of iteration 3 of the first problem
    // of block 0, left in the unused words of the control block.  No stamp exists in the shipped build.
#ifdef GBDPCG_CL_STAMPS
#define GBDPCG_CL_STAMP(IDX, WAVE, COND)                                                                             \
    if (blk == 0 && wave == (WAVE) && lane == 0 && (COND)) reinterpret_cast<u64 *>(ws)[IDX] = __builtin_amdgcn_s_memtime();
// ... and the 100 MHz real-time clock (wall time, whatever the shader clock does) for the problem-level stamps
#define GBDPCG_CL_STAMP_RT(IDX, WAVE, COND)                                                                          \
    if (blk == 0 && wave == (WAVE) && lane == 0 && (COND)) reinterpret_cast<u64 *>(ws)[IDX] = __builtin_amdgcn_s_memrealtime();
// start (0) / end (1) of every workgroup, behind the slots as sized for the device (256 CUs on the MI355X)
#define GBDPCG_CL_STAMP_WG(WHICH)                                                                                    \
    if (tid == 0) reinterpret_cast<u64 *>(ws + kClCtrlBytes + 2u * 256u * kClSlotBytes)[2 * blk + (WHICH)] = __builtin_amdgcn_s_memrealtime();
#else
#define GBDPCG_CL_STAMP_WG(WHICH)
#define GBDPCG_CL_STAMP(IDX, WAVE, COND)
#define GBDPCG_CL_STAMP_RT(IDX, WAVE, COND)
#endif

    // this launch's number, in the tag bits above the epoch (epochs of a launch fit epoch_bits: the host checked)
    const uint32_t nonce = (uint32_t)(__hip_atomic_load(reinterpret_cast<u64 *>(ws) + 31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % kClLaunchMod + 1ull) << epoch_bits;
    bool greeted = false;     // HELLO done
    bool same_xcd = false;    // every member of the cluster runs on this XCD: publish with plain stores (set by HELLO)
    bool dead = false;        // a hand-off of this cluster timed out: its remaining problems are left to the member that leaves last
    uint32_t first_dead = 0xffffffffu;   // ... from this problem on
    // problems of this cluster SOLVED BY THIS LAUNCH so far: the epochs of a problem continue where the last one stopped.
    // Problems another launch owns do not count: the first epochs a launch polls for must be 1 and 2 whatever the batch
    // holds.
    uint32_t ordinal = 0;
    const uint32_t epochs_per_problem = 2u * a.max_iter + 4u;
    uint32_t xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    xcc_id = (xcc_id & 0xfu) + 1u;   // never 0: a payload of its own kind

    GBDPCG_CL_STAMP_RT(21, 0, true)
    GBDPCG_CL_STAMP_WG(0)
    // Does this launch own ANY problem of this cluster?  In the default (tested) symmetric mode it usually owns none, and the
    // launch should cost as little as possible: the verdicts of 64 problems are fetched in one round trip, instead of one
    // round trip per problem in the loop below.  (Using the masks inside the loop too made its iterations 2 % slower:
    // register allocation.)
    bool any = a.sel == nullptr;
    for (uint32_t first = c; !any && first < a.batch; first += 64 * clusters) {
        const uint32_t left = (a.batch - first + clusters - 1) / clusters;
        any = pcg_takes_mask(a, first, clusters, left < 64 ? left : 64u, lane) != 0ull;
    }
    if (any) {
        // this workgroup's slot, both parities, cleared by the lanes that will publish into the same words (a wave's stores
        // to one address arrive in order): tag 0 is nobody's
        const cl_u32x2 z2 = {0u, 0u};
        const cl_u32x4 z4 = {0u, 0u, 0u, 0u};
        uint32_t lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (uint32_t par = 0; par < 2; ++par) {
            const uint32_t o = par * par_stride + my_slot;
            if (lo == 0) __builtin_amdgcn_raw_buffer_store_b64(z2, region, (int)(o + wave * 8), 0, kClSc1);
            if (wave == 0 && has_left && lo < n / 2) __builtin_amdgcn_raw_buffer_store_b128(z4, region, (int)(o + kClFirstOff + lo * 16), 0, kClSc1);
            if (wave == wl && has_right && lo - lb * (n / 2) < n / 2)
                __builtin_amdgcn_raw_buffer_store_b128(z4, region, (int)(o + kClLastOff + (lo - lb * (n / 2)) * 16), 0, kClSc1);
        }
    }
    if (any) for (uint32_t prob = c; prob < a.batch; prob += clusters) {
        if (!pcg_takes(a, prob)) continue;   // this launch is not the one that owns the problem (same verdict in every member)
        if (dead) {
            if (a.rescue_off && h == 0 && tid == 0) {   // (hooks build only: show the mark)
                a.iters[prob] = kItersGaveUp;
                if (a.max_iter_exit) a.max_iter_exit[prob] = 2;
            }
            continue;
        }
        GBDPCG_CL_STAMP_RT(14, 0, ordinal == 3)
        float total = 0.f, g0 = 0.f, g1 = 0.f, h0 = 0.f, h1 = 0.f, part;
        uint32_t hidx = 0xffffffffu;
        bool ok = true;
        if (!greeted) {
            // HELLO (epoch 1): where does everybody run?  The payload of every partial granule is the sender's XCD id.
            greeted = true;
            const float my_id = __builtin_bit_cast(float, xcc_id);
            if (tid == 0) bci[0] = 0u;
            wg_barrier();
            GBDPCG_CL_HANDOFF(1u, my_id, 0.f, 0.f, xa, total, g0, g1, h0, h1, hidx, ok)
            if (wave == POLL) {
                uint32_t lo = lane;
                asm volatile("" : "+v"(lo));
                const bool mine = lo >= 4 * H || (fbits(g0) == xcc_id && fbits(g1) == xcc_id);
                const bool all_here = __all(mine);
                if (lane == 0) {
                    bci[0] = ok ? 0u : 2u;
                    bci[1] = ok && all_here ? 1u : 0u;
                }
            }
            wg_barrier();
            same_xcd = bci[1] != 0u && !no_plain;
            if (bci[0] == 2u) {   // the cluster never got together: nothing of it is solved here
                dead = true;
                first_dead = prob;
                if (a.rescue_off && h == 0 && tid == 0) {
                    a.iters[prob] = kItersGaveUp;
                    if (a.max_iter_exit) a.max_iter_exit[prob] = 2;
                }
                wg_barrier();
                continue;
            }
        }
        const uint32_t e0 = 1u + ordinal * epochs_per_problem;   // epochs e0 + 1 .. e0 + 2 max_iter + 2 belong to this problem
        const float *S = a.S + prob * mstride;
        const float *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;   // nullptr: identity preconditioner
        const size_t voff = (size_t)prob * len;

        // lambda and gamma of this lane's rows and the two halo knots of lambda (straight from the input vector: no member
        // writes lambda before every member has passed its first hand-off) are requested next to the first tile stages, so
        // that their round trips run under the 300 KB of tile loads instead of in front of the first product
        float lamv[V], gamv[V], halo_lam = 0.f;
        auto request_vectors = [&]() {
            uint32_t lo = lane;
            asm volatile("" : "+v"(lo));   // addresses from the lane number, not from registers held since the kernel started
            const size_t g0 = voff + (size_t)k_lo * n + wave * (Dg::BPW * n) + 2 * lo;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                lamv[j] = dc.live ? a.lambda[g0 + j] : 0.f;
                gamv[j] = dc.live ? a.gamma[g0 + j] : 0.f;
            }
            const uint32_t t = wave * 64 + lo;
            if (t < 2 * n) {   // threads [0, n): the knot before the own ones, [n, 2n): the knot after them
                const int64_t gi = t < n ? (int64_t)k_lo * n - n + t : (int64_t)(k_lo + cnt) * n + (t - n);
                if (gi >= 0 && gi < (int64_t)len) halo_lam = a.lambda[voff + gi];
            }
        };

        DenseTile<float, NCT, V> tS, tP;
        if constexpr (STAGED) {
            GBDPCG_CL_STAMP(12, 0, ordinal == 0)
            dense_staged_load<NCT, V>(S, P, N, dc, wave, lane, k_lo, cnt, reinterpret_cast<float *>(stage_raw), tS, tP,
                                      request_vectors);   // the vectors are requested behind the first two tile stages
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last picks are in registers before anything else happens
            GBDPCG_CL_STAMP(13, 0, ordinal == 0)
            GBDPCG_CL_STAMP_RT(15, 0, ordinal == 3)
        } else {
            request_vectors();
            dense_load<float, NCT, V>(S, N, dc, tS);
            if (P) {
                dense_load<float, NCT, V>(P, N, dc, tP);
            } else {
#pragma unroll
                for (uint32_t cc = 0; cc < Dg::COLS; ++cc) tP.a[cc][0] = tP.a[cc][1] = 0.f;
            }
        }

        float rv[V], pv[V], yv[V];
        // windows: lambda on the own knots and the halos, zeros behind them (rows past the own knots, halos at the ends
        // of the problem); every entry has exactly one writer
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xa[n + row0 + j] = lamv[j];
        }
        if (tid < 2 * n) {
            const uint32_t i = tid < n ? tid : (cnt + 1) * n + (tid - n);
            xa[i] = halo_lam;
            xb[i] = 0.f;
        }
        {
            uint32_t t = tid;
            asm volatile("" : "+v"(t));
            for (uint32_t i = (cnt + 2) * n + t; i < WINF; i += THREADS) xa[i] = xb[i] = 0.f;
        }
        if (tid == 0) bci[0] = 0u;
        wg_barrier();
        if (PTAIL > 0 && P) {   // (behind the barrier: every wave is done with its staging buffers, which this overwrites)
#pragma unroll
            for (int t = 0; t < PTAIL; ++t)
                ptail[t * THREADS + tid] = make_float2(tP.a[Dg::COLS - PTAIL + t][0], tP.a[Dg::COLS - PTAIL + t][1]);
        }
        GBDPCG_CL_STAMP_RT(16, 0, ordinal == 3)

        // r = gamma - S lambda                                            (pcg.cuh:118-126); the boundary knots of r travel
        dense_mv<float, NCT, V, GBDPCG_CL_CHAINS>(tS, xa, dc, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) rv[j] = dc.live ? gamv[j] - yv[j] : 0.f;
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xb[n + row0 + j] = rv[j];
        }
        GBDPCG_CL_HANDOFF(e0 + 1u, 0.f, rv[0], rv[1], xb, total, g0, g1, h0, h1, hidx, ok)
        if (wave == POLL) {
            if (!ok && lane == 0) bci[0] = 2u;
            if (ok && hidx != 0xffffffffu) { xb[hidx] = g0; xb[hidx + 1] = g1; }
        }
        wg_barrier();
        bool failed = bci[0] == 2u;
        GBDPCG_CL_STAMP_RT(17, 0, ordinal == 3)

        // r~ = Pinv r ; p = r~ ; eta = r.r~                               (pcg.cuh:130-149)
        float eta = 0.f;
        if (!failed) {
            if (P) dense_mv<float, NCT, V, GBDPCG_CL_CHAINS, PTAIL>(tP, xb, dc, yv, ptail + tid, THREADS);
            part = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                pv[j] = P ? yv[j] : rv[j];
                part = fma_t(rv[j], pv[j], part);
            }
            part = wave_sum(part);
            if (dc.live) {   // every read of xa (as lambda) happened before the last barrier
#pragma unroll
                for (int j = 0; j < V; ++j) xa[n + row0 + j] = pv[j];
            }
            GBDPCG_CL_HANDOFF(e0 + 2u, part, pv[0], pv[1], xa, total, g0, g1, h0, h1, hidx, ok)
            if (wave == POLL) {
                if (lane == 0) {
                    if (!ok) bci[0] = 2u;
                    bc[0] = total;
                }
                if (ok && hidx != 0xffffffffu) { xa[hidx] = g0; xa[hidx + 1] = g1; }
            }
            wg_barrier();
            failed = bci[0] == 2u;
            eta = bc[0];
        }

        GBDPCG_CL_STAMP_RT(18, 0, ordinal == 3)
        uint32_t iter = 0;
        bool max_iter_exit = true;
        for (; !failed && iter < a.max_iter; ++iter) {                    // pcg.cuh:154
            // upsilon = S p ; alpha = eta / (p.upsilon)                   (pcg.cuh:156-169)
            GBDPCG_CL_STAMP(1, POLL, iter == 3 && ordinal == 0)
            GBDPCG_CL_STAMP(8, 0, iter == 3 && ordinal == 0)
            dense_mv<float, NCT, V, GBDPCG_CL_CHAINS>(tS, xa, dc, yv);
            part = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) part = fma_t(pv[j], yv[j], part);
            part = wave_sum(part);
            GBDPCG_CL_STAMP(2, POLL, iter == 3 && ordinal == 0)
            GBDPCG_CL_STAMP(9, 0, iter == 3 && ordinal == 0)
            GBDPCG_CL_HANDOFF(e0 + 3u + 2u * iter, part, yv[0], yv[1], xb, total, g0, g1, h0, h1, hidx, ok)
            GBDPCG_CL_STAMP(3, POLL, iter == 3 && ordinal == 0)
            GBDPCG_CL_STAMP(10, 0, iter == 3 && ordinal == 0)
            if (wave == POLL) {
                const float al = eta / total;
                if (lane == 0) {
                    if (!ok) bci[0] = 2u;
                    bc[0] = al;
                }
                // r -= alpha upsilon on the halo knots: the owner's fma on the owner's bits
                if (ok && hidx != 0xffffffffu) {
                    xb[hidx] = fma_t(-al, g0, h0);
                    xb[hidx + 1] = fma_t(-al, g1, h1);
                }
            }
            wg_barrier();
            GBDPCG_CL_STAMP(4, POLL, iter == 3 && ordinal == 0)
            if (bci[0] == 2u) { failed = true; break; }
            const float alpha = bc[0];
            // lambda += alpha p ; r -= alpha upsilon                      (pcg.cuh:172-176)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                lamv[j] = fma_t(alpha, pv[j], lamv[j]);
                rv[j] = fma_t(-alpha, yv[j], rv[j]);
            }
            if (dc.live) *reinterpret_cast<float2 *>(xb + own_idx()) = make_float2(rv[0], rv[1]);
            wg_barrier();
            GBDPCG_CL_STAMP(5, POLL, iter == 3 && ordinal == 0)
            // r~ = Pinv r ; eta_new = r.r~                                (pcg.cuh:180-193)
            if (P) dense_mv<float, NCT, V, GBDPCG_CL_CHAINS, PTAIL>(tP, xb, dc, yv, ptail + tid, THREADS);
            part = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (!P) yv[j] = rv[j];
                part = fma_t(rv[j], yv[j], part);
            }
            part = wave_sum(part);
            GBDPCG_CL_HANDOFF(e0 + 4u + 2u * iter, part, yv[0], yv[1], xa, total, g0, g1, h0, h1, hidx, ok)
            if (wave == POLL) {
                const bool conv = fabsf(total) < a.tol;                   // pcg.cuh:195
                const float be = total / eta;                             // pcg.cuh:199
                if (lane == 0) {
                    bci[0] = !ok ? 2u : (conv ? 1u : 0u);
                    bc[0] = total;
                    bc[1] = be;
                }
                // p = r~ + beta p on the halo knots                       (pcg.cuh:203-206)
                if (ok && !conv && hidx != 0xffffffffu) {
                    xa[hidx] = fma_t(be, h0, g0);
                    xa[hidx + 1] = fma_t(be, h1, g1);
                }
            }
            wg_barrier();
            const uint32_t verdict = bci[0];
            if (verdict == 2u) { failed = true; break; }
            if (verdict == 1u) {
                ++iter;
                max_iter_exit = false;
                break;
            }
            GBDPCG_CL_STAMP(6, POLL, iter == 3 && ordinal == 0)
            const float beta = bc[1];
            eta = bc[0];
#pragma unroll
            for (int j = 0; j < V; ++j) pv[j] = fma_t(beta, pv[j], yv[j]);
            if (dc.live) *reinterpret_cast<float2 *>(xa + own_idx()) = make_float2(pv[0], pv[1]);
            wg_barrier();
            GBDPCG_CL_STAMP(7, POLL, iter == 3 && ordinal == 0)
        }

        GBDPCG_CL_STAMP_RT(19, 0, ordinal == 3)
        // outputs                                                         (pcg.cuh:212,215)
        if (dc.live && !failed) {
            // (the row index from an opaque copy of the lane number: as three 64-bit addresses computed in front of the problem
            // loop these were the kernel's last spills to scratch)
            uint32_t lo = lane;
            asm volatile("" : "+v"(lo));
            const size_t g = voff + (size_t)k_lo * n + wave * (Dg::BPW * n) + 2 * lo;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                a.lambda[g + j] = lamv[j];
                if (a.r) a.r[g + j] = rv[j];
                if (a.p) a.p[g + j] = pv[j];
            }
        }
        if (h == 0 && tid == 0 && (!failed || a.rescue_off)) {
            a.iters[prob] = failed ? kItersGaveUp : iter;
            if (a.max_iter_exit) a.max_iter_exit[prob] = failed ? 2 : (max_iter_exit ? 1 : 0);
        }
        dead = failed;
        if (failed) first_dead = prob;
        wg_barrier();   // the windows and bci are reused by the next problem
        GBDPCG_CL_STAMP_RT(20, 0, ordinal == 3)
        ++ordinal;
    }

    GBDPCG_CL_STAMP_RT(22, 0, true)
    // ---- a cluster that could not meet: the member that leaves last solves what was given up, alone -------------------------
    // Every member that gave up did so on the same problem (nobody gets past a problem without everybody's hand-offs; the
    // one exception -- a member stalled for the whole spin bound exactly at a problem's last hand-off, which then finishes
    // that problem while its partner has given it up -- is why the members report the problem and the smallest one wins:
    // that problem is then solved again, by the streaming kernel, to the same tolerance).  The counter is bumped after the
    // report (s_waitcnt in between), both with agent-scope atomics; the last member puts both words back.
    if (any && !a.rescue_off && a.rescue_vec) {
        u64 *left = reinterpret_cast<u64 *>(ws + kClLeftOff) + 2 * c;
        if (tid == 0) {
            if (dead) __hip_atomic_fetch_max(left + 1, (u64)(0xffffffffu - first_dead), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const u64 before = __hip_atomic_fetch_add(left, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t from = 0xffffffffu;
            if (before == H - 1u) {
                const u64 worst = __hip_atomic_exchange(left + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(left, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (worst != 0ull) from = 0xffffffffu - (uint32_t)worst;
            }
            bci[2] = from;
        }
        __syncthreads();
        const uint32_t from = bci[2];
        if (from != 0xffffffffu) {
            float *vec = reinterpret_cast<float *>(a.rescue_vec) + (size_t)c * rescue_vec_elems<float>(n, N);
            for (uint32_t prob = from; prob < a.batch; prob += clusters)
                if (pcg_takes(a, prob)) stream_rescue<float, Dg::WAVES>(a, prob, vec, rescue_red);
        }
    }
    // ---- the workgroup that finishes last gives the next launch its number ------------------------------------------------
    // (every workgroup of this launch has read the number by then; the counter and the number are only ever touched with
    // agent-scope atomics)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        u64 *ctl = reinterpret_cast<u64 *>(ws);
        const u64 before = __hip_atomic_fetch_add(ctl + 30, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == grid - 1u) {
            __hip_atomic_store(ctl + 30, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ctl + 31, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GBDPCG_CL_STAMP_WG(1)
#undef GBDPCG_CL_HANDOFF
#undef GBDPCG_CL_STAMP
#undef GBDPCG_CL_STAMP_WG
#undef GBDPCG_CL_STAMP_RT
}

// The block sizes the kernel is built for (fp32, two rows per lane, n / 2 <= 8 lanes per knot; the staged tile loads want
// n^2 % 4 == 0).  2 x 3n x 2 matrix registers per lane must leave room for the working set: at n = 16 (192) the R block of
// Pinv stays in LDS (ClusterTail).
#define GBDPCG_CLUSTER_N(X) X(8) X(10) X(12) X(14) X(16)

// fp32, general storage, horizons beyond what ONE workgroup keeps in registers (pcg_resident.hip: 8 waves x floor(64 / (n/2))
// knots -- 72 at n = 14, 80 at n = 12) up to kClMaxH times that.  GBDPCG_NO_CLUSTER disables the path (tuning runs).
template <typename T> uint32_t cluster_members(uint32_t n, uint32_t N)
{
    static const bool off = getenv("GBDPCG_NO_CLUSTER") != nullptr;
    if (off || sizeof(T) != 4) return 0;
    uint32_t per_wg = 0;
#define GBDPCG_X(NN) \
    if (n == NN) per_wg = DenseGeom<float, NN, 2>::MAX_KNOTS;
    GBDPCG_CLUSTER_N(GBDPCG_X)
#undef GBDPCG_X
    if (per_wg == 0 || N <= per_wg) return 0;   // not built for the block size / pcg_resident.hip has it in one workgroup
    const uint32_t H = (N + per_wg - 1) / per_wg;
    return H <= kClMaxH ? H : 0;
}

// ... plus 16 bytes per CU behind the slots: start / end of every workgroup on the real-time clock (diagnostic build only)
size_t cluster_workspace_bytes(const DeviceInfo &dev) { return kClCtrlBytes + (size_t)2 * dev.num_cus * kClSlotBytes + (size_t)dev.num_cus * 16; }

// Device memory for the in-kernel rescue: one set of vectors per cluster, sized for the longest horizon the path takes.
size_t cluster_rescue_bytes(const DeviceInfo &dev)
{
    size_t elems = 0;
#define GBDPCG_X(NN)                                                                                       \
    {                                                                                                      \
        const size_t e = rescue_vec_elems<float>(NN, kClMaxH * DenseGeom<float, NN, 2>::MAX_KNOTS);        \
        elems = e > elems ? e : elems;                                                                     \
    }
    GBDPCG_CLUSTER_N(GBDPCG_X)
#undef GBDPCG_X
    return (size_t)(dev.num_cus / 2) * elems * sizeof(float);
}

template <typename T>
bool launch_pcg_cluster(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err)
{
    if constexpr (sizeof(T) == 4) {
        const uint32_t H = cluster_members<T>(a.n, a.N);
        if (H == 0 || !a.cluster_ws || a.symmetric) return false;
        if ((reinterpret_cast<uintptr_t>(a.S) % 8) || (a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % 8)) return false;
        uint32_t clusters = (uint32_t)dev.num_cus / H;   // one workgroup per CU: a cluster's members are resident together
        if (clusters > a.batch) clusters = a.batch;
        if (clusters == 0) return false;
        const uint32_t rounds = (a.batch + clusters - 1) / clusters;
        // a tag = {launch number, epoch}: the epochs of a launch take the low epoch_bits of its 32 bits
        const double epochs = 2.0 + (double)rounds * (2.0 * a.max_iter + 4.0);
        if (epochs >= (double)(1u << kClEpochBits)) return false;
        const uint32_t C = (a.N + H - 1) / H;
        uint32_t spin_limit = 1u << 21;      // polls before a hand-off is given up (~1 us per poll: about two seconds)
        uint32_t drop_block = 0xffffffffu;
#ifdef GBDPCG_TEST_HOOKS
        // variants/libgbdpcg_hooks.so only (tests/test_gpu_cluster.py): a short bound, and a workgroup that never publishes
        if (const char *e = getenv("GBDPCG_CLUSTER_SPIN_LIMIT")) spin_limit = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBDPCG_CLUSTER_DROP_WG")) drop_block = (uint32_t)strtoul(e, nullptr, 10);
        const bool rescue_off = getenv("GBDPCG_RESCUE_OFF") != nullptr;   // show a test what a cluster that gave up leaves behind
#else
        const bool rescue_off = false;
#endif
        PcgArgs<T> ka = a;
        ka.rescue_off = rescue_off;
        // coalesced LDS-DMA tile loads need 16-byte aligned matrices (every hipMalloc'ed buffer is)
        static const bool no_staging = getenv("GBDPCG_CLUSTER_DIRECT_LOADS") != nullptr;   // tuning runs only
        const bool staged = !no_staging && !((reinterpret_cast<uintptr_t>(a.S) | reinterpret_cast<uintptr_t>(a.Pinv)) % 16);
        static const bool no_plain = getenv("GBDPCG_CLUSTER_NO_PLAIN") != nullptr;   // tuning runs: always sc1 stores
        bool launched = false;
#define GBDPCG_X(NN)                                                                                                     \
        if (a.n == NN) {                                                                                                 \
            auto kern = staged ? pcg_cluster_kernel<NN, 2, true> : pcg_cluster_kernel<NN, 2, false>;                     \
            const size_t tail = (size_t)ClusterTail<NN>::COLS * 512 * sizeof(float2);                                   \
            const size_t lds = staged ? (dense_stage_lds_bytes<NN, 2>() > tail ? dense_stage_lds_bytes<NN, 2>() : tail) : tail; \
            /* on every launch, like the other launchers: HIP keeps the attribute per device */                          \
            if (lds) {                                                                                                   \
                *err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (*err != hipSuccess) return true;                                                                     \
            }                                                                                                            \
            hipLaunchKernelGGL(kern, dim3(clusters * H), dim3(512), lds, s, ka, static_cast<unsigned char *>(a.cluster_ws), H, C, \
                               clusters, spin_limit, drop_block, no_plain);                                              \
            launched = true;                                                                                             \
        }
        GBDPCG_CLUSTER_N(GBDPCG_X)
#undef GBDPCG_X
        if (!launched) return false;
        *err = hipGetLastError();
        return true;
    } else {
        return false;
    }
}

template uint32_t cluster_members<float>(uint32_t, uint32_t);
template uint32_t cluster_members<double>(uint32_t, uint32_t);
template bool launch_pcg_cluster<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t, hipError_t *);
template bool launch_pcg_cluster<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t, hipError_t *);

}  // namespace gbdpcg
