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
// members (0: none) | slots[2 parities][grid blocks]].  One slot:
//   +0    the 8 wave partials (fp32: 8 granules of 8 bytes; fp64: 8 pairs of granules, 128 bytes)
//   +128  the first own knot of the product vector, for the left neighbour: one 16-byte store {w0, tag, w1, tag} per lane of
//         the knot (the lane's 8 bytes of payload: two fp32 rows, one fp64 row, or one fp32 row), up to 16 lanes
//   +384  the last own knot, for the right neighbour
constexpr uint32_t kClLeftOff = 256, kClCtrlBytes = 256 + 256 * 16, kClSlotBytes = 640, kClFirstOff = 128, kClLastOff = 384;
// members of a cluster at most: the polling wave gives PPM lanes to every member's partials (fp32: 4, fp64: 8) and has 32 for them
// (round 3: eight in fp32 -- 1024 converged solves of 14 x 300: 1,897 us streamed -> 997 us, one problem of 14 x 512: 85 -> 59 us)
template <typename T> constexpr uint32_t cl_max_members() { return sizeof(T) == 4 ? 8u : 4u; }
constexpr uint32_t kClEpochBits = 20, kClLaunchMod = 4095;   // tag = ((launch mod 4095) + 1) << 20 | epoch


// Lane roles of the polling wave in one gather: lanes [0, PPM H) fetch 16 bytes of wave partials each (fp32: PPM = 4 lanes per
// member, two partials per lane; fp64: PPM = 8, one), lanes [32, 32 + LPB) the left neighbour's last knot, lanes
// [48, 48 + LPB) the right neighbour's first knot (LPB = lanes per knot <= 16).
constexpr uint32_t kClLeftLane = 32, kClRightLane = 48;

// A lane's payload in a boundary store: its V rows as two 32-bit words.
template <typename T, int V> __device__ __forceinline__ void cl_pack(const T (&v)[V], uint32_t &w0, uint32_t &w1)
{
    if constexpr (sizeof(T) == 8) {
        const u64 b = __builtin_bit_cast(u64, v[0]);
        w0 = (uint32_t)b;
        w1 = (uint32_t)(b >> 32);
    } else {
        w0 = __builtin_bit_cast(uint32_t, v[0]);
        w1 = V > 1 ? __builtin_bit_cast(uint32_t, v[V - 1]) : 0u;
    }
}
template <typename T, int V> __device__ __forceinline__ void cl_unpack(uint32_t w0, uint32_t w1, T (&v)[V])
{
    if constexpr (sizeof(T) == 8) {
        v[0] = __builtin_bit_cast(double, ((u64)w1 << 32) | w0);
    } else {
        v[0] = __builtin_bit_cast(float, w0);
        if constexpr (V > 1) v[1] = __builtin_bit_cast(float, w1);
    }
}
// The wave partials a partial lane fetched (16 bytes): two fp32 partials, or one fp64 partial.
template <typename T> __device__ __forceinline__ T cl_partials(uint32_t w0, uint32_t w1)
{
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(double, ((u64)w1 << 32) | w0);
    else return __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
}

#ifndef GBDPCG_CL_PTAIL
#define GBDPCG_CL_PTAIL 0   // columns of a lane's Pinv block-row kept in LDS instead of registers (0, 2, 4, ...): not needed since the kernel compiles without scratch; kept for A/B builds
#endif
template <typename T, int NCT> struct ClusterTail {
    static constexpr int COLS = (sizeof(T) == 4 && NCT == 18) ? 36 : (sizeof(T) == 8 && NCT == 16) ? 32 : (sizeof(T) == 4 && NCT == 16) ? 16 : (sizeof(T) == 8 && NCT == 14) ? 14 : (NCT % 2 == 0 ? GBDPCG_CL_PTAIL : 0);
};
// bytes between the tails of two waves in dynamic LDS
template <typename T, int NCT, int V, bool STAGED> struct ClusterTailBytes {
    static constexpr size_t PER_WAVE = STAGED ? (size_t)DenseStage<T, NCT, V>::BYTES : (size_t)ClusterTail<T, NCT>::COLS * 64 * 8;
};
#ifndef GBDPCG_CL_CHAINS
#define GBDPCG_CL_CHAINS 3   // accumulator chains of a block-row product (bt_dense.hpp, dense_mv); 1 for A/B builds
#endif

}  // namespace

// Instantiations light enough for two workgroups to share a compute unit (launch_pcg_cluster): held to 128 registers.  fp32 at
// stateSize 2, 3, 4, 5, 7, 9, 11 need 108-118 anyway; 6 (144), 13 (132) and fp64 at 2-5 (139) are asked to fit (2-12 registers
// spilled, and still 1.3-1.6x faster with two per CU: 1024 converged solves of 9 x 128: 516 -> 317 us, 13 x 128: 491 -> 367,
// 6 x 200: 212 -> 151, fp64 4 x 200: 208 -> 134).  stateSize 15 (143: 20 spilled) gains 14 % that way (531 -> 455 us) and never runs as
// a cluster of one, where the cap alone would cost 30 %: pcg_resident.hip has its short horizons.
template <typename T, int NCT> struct ClusterLight {
    static constexpr bool value = (sizeof(T) == 4 && (NCT <= 7 || NCT == 9 || NCT == 11 || NCT == 13 || NCT == 15)) || (sizeof(T) == 8 && NCT <= 5);
};
// ONE: a cluster of one workgroup (launch_pcg_cluster: only where the shape has such launches) -- nobody to hand anything to, the wave
// partials meet in LDS, in the same words and the same sum as in memory.  A compile-time switch: as a run-time test inside the
// hand-off it cost the clusters of two to four members 4 %.
template <typename T, int NCT, int V, bool STAGED, bool ONE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(ClusterLight<T, NCT>::value ? 4 : 2)))
void pcg_cluster_kernel(PcgArgs<T> a, unsigned char *ws, uint32_t H, uint32_t C,
                                                          uint32_t clusters, uint32_t spin_limit, uint32_t drop_block, bool no_plain)
{
#ifndef GBDPCG_TEST_HOOKS
    drop_block = 0xffffffffu;   // the hook that silences one workgroup exists in variants/libgbdpcg_hooks.so only
#endif
    constexpr uint32_t epoch_bits = kClEpochBits;
    using Dg = DenseGeom<T, NCT, V>;
    // lane roles below: LPB = n / V lanes per knot (<= 16: a knot's boundary values are LPB 16-byte stores), a lane's rows are 8
    // bytes of payload at most, 8 waves (partial lanes: 16 bytes of the 8 H wave partials each)
    static_assert(V * sizeof(T) <= 8 && NCT % V == 0 && Dg::LPB <= 16 && Dg::WAVES == 8, "lane roles of the hand-off");
    constexpr uint32_t n = NCT, LPB = Dg::LPB, THREADS = Dg::WAVES * 64, WINF = align16<T>((Dg::MAX_KNOTS + 2) * n);
    constexpr uint32_t PPM = sizeof(T) == 4 ? 4u : 8u;   // partial lanes per member
    constexpr int CHAINS = NCT % 2 == 0 ? GBDPCG_CL_CHAINS : 1;   // (one chain per block needs whole operand pairs per block)
    __shared__ __attribute__((aligned(16))) T xa[WINF];   // window of p (lambda in the prologue): halo knot, own knots, halo knot
    __shared__ __attribute__((aligned(16))) T xb[WINF];   // window of r
    __shared__ T bc[4];           // [0] alpha / eta' of the phase just gathered, [1] beta
    __shared__ uint32_t bci[4];   // [0] 0 go on, 1 converged, 2 hand-off timed out; [2], [3] the rescue's bookkeeping
    __shared__ T rescue_red[2 * Dg::WAVES];
    __shared__ cl_u32x4 lpart[ONE ? Dg::WAVES : 1];   // ONE: the wave partials of a hand-off, laid out like the slot in memory
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];   // STAGED: dense_stage_lds_bytes (bt_dense.hpp)
    // The last PTAIL columns of this lane's block-row of Pinv (its V rows each) live in LDS instead of registers (dense_mv, TAIL):
    // at n = 16 the whole R block of Pinv -- 2 x 96 matrix registers per lane do not fit next to the working set, 96 + 64 do;
    // likewise in fp64 at n = 14 (2 x 84 registers of matrix data next to a working set of twice the width: 84 + 56).
    // They take the place of the staging buffers once every wave's tiles are in (cluster_lds_bytes).
    constexpr int PTAIL = ClusterTail<T, NCT>::COLS;
    // Every wave keeps its own: in its first staging buffer when the tiles come in staged (bt_dense.hpp, dense_staged_load: the
    // block goes from the second buffer straight there, never through registers), else in PTAIL x 64 elements of its own.
    using TV = typename DenseTailElem<T, V>::type;
    static_assert(PTAIL == 0 || sizeof(TV) == 8, "a lane's rows of a tail column are 8 bytes");

    const uint32_t N = a.N, len = n * N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TV *ptail = reinterpret_cast<TV *>(stage_raw + (size_t)wave * ClusterTailBytes<T, NCT, V, STAGED>::PER_WAVE);
    // PTAIL <= n: columns of the R block, behind pt_lane.  PTAIL = 2n (stateSize 18): the D block behind ptd_lane (staged: re-laid inside
    // the wave's first staging buffer) and the R block behind pt_lane (staged: inside its second buffer).
    constexpr bool TWO = PTAIL > (int)NCT;
    static_assert(!TWO || PTAIL == 2 * (int)NCT, "the D and the R block, whole");
    const TV *ptd_lane = ptail + lane;
    const TV *pt_lane = !TWO ? ptail + lane
                        : STAGED ? reinterpret_cast<const TV *>(stage_raw + (size_t)(Dg::WAVES + wave) * ClusterTailBytes<T, NCT, V, STAGED>::PER_WAVE) + lane
                                 : ptail + NCT * 64 + lane;
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

    const DenseCtx<T, NCT, V> dc(wave, lane, cnt, k_lo);
    // first of this lane's rows, counted from knot k_lo: (wave BPW + lane / LPB) n + (lane % LPB) V = wave BPW n + V lane (n = V LPB)
    const uint32_t row0 = dc.live ? wave * (Dg::BPW * n) + V * lane : 0u;
    // the same index inside a window, from an opaque copy of the lane number (see GBDPCG_CL_HANDOFF: not worth a register)
    auto own_idx = [&]() {
        uint32_t lo = lane;
        asm volatile("" : "+v"(lo));
        return n + wave * (Dg::BPW * n) + V * lo;
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
#define GBDPCG_CL_HANDOFF(EPOCH, PART, VALS, HWIN, TOTAL, GV, HV, HIDX, OK, GOTX, GOTZ)                               \
    {                                                                                                                \
        const uint32_t tag = nonce | (EPOCH), par_off = ((EPOCH) & 1u) * par_stride;                                 \
        uint32_t lo = lane;                                                                                          \
        asm volatile("" : "+v"(lo));                                                                                 \
        if (blk != drop_block) {                                                                                     \
            uint32_t w0, w1, q0, q1;                                                                                 \
            cl_pack<T, V>(VALS, w0, w1);                                                                             \
            const T part_[1] = {PART};                                                                               \
            cl_pack<T, 1>(part_, q0, q1);                                                                            \
            const cl_u32x2 x2 = {q0, tag};                                                                           \
            const cl_u32x4 x4 = {q0, tag, q1, tag};                                                                  \
            const cl_u32x4 bx = {w0, tag, w1, tag};                                                                  \
            const int o_part = (int)(par_off + my_slot + wave * (sizeof(T) == 4 ? 8u : 16u)),                        \
                      o_first = (int)(par_off + my_slot + kClFirstOff + lo * 16),                                    \
                      o_last = (int)(par_off + my_slot + kClLastOff + (lo - lb * LPB) * 16);                         \
            const bool p_first = wave == 0 && has_left && lo < LPB, p_last = wave == wl && has_right && lo - lb * LPB < LPB; \
            if constexpr (ONE) {                                                                                     \
                if (lo == 0) {                                                                                       \
                    if constexpr (sizeof(T) == 4) reinterpret_cast<cl_u32x2 *>(lpart)[wave] = x2;                    \
                    else lpart[wave] = x4;                                                                           \
                }                                                                                                    \
            } else if (same_xcd) {   /* plain: the line stays in this XCD's L2, where every member of the cluster polls */   \
                if (lo == 0) {                                                                                       \
                    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b64(x2, region, o_part, 0, 0);   \
                    else __builtin_amdgcn_raw_buffer_store_b128(x4, region, o_part, 0, 0);                           \
                }                                                                                                    \
                if (p_first) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_first, 0, 0);                      \
                if (p_last) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_last, 0, 0);                        \
            } else {          /* sc1: write-through, seen from any XCD */                                            \
                if (lo == 0) {                                                                                       \
                    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b64(x2, region, o_part, 0, kClSc1); \
                    else __builtin_amdgcn_raw_buffer_store_b128(x4, region, o_part, 0, kClSc1);                      \
                }                                                                                                    \
                if (p_first) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_first, 0, kClSc1);                 \
                if (p_last) __builtin_amdgcn_raw_buffer_store_b128(bx, region, o_last, 0, kClSc1);                   \
            }                                                                                                        \
        }                                                                                                            \
        if constexpr (ONE) wg_barrier();                                                                             \
        if (wave == POLL) {                                                                                          \
            uint32_t poll_off = 0;                                                                                   \
            bool have = true;                                                                                        \
            HIDX = 0xffffffffu;                                                                                      \
            if (lo < PPM * H) {                                                                                      \
                poll_off = (blk0 + (lo / PPM) * bstride) * kClSlotBytes + (lo % PPM) * 16;                           \
                have = false;                                                                                        \
            } else if (lo - kClLeftLane < LPB && has_left) {                                                         \
                poll_off = (blk - bstride) * kClSlotBytes + kClLastOff + (lo - kClLeftLane) * 16;                    \
                HIDX = (lo - kClLeftLane) * V;                                                                       \
                have = false;                                                                                        \
            } else if (lo - kClRightLane < LPB && has_right) {                                                       \
                poll_off = (blk + bstride) * kClSlotBytes + kClFirstOff + (lo - kClRightLane) * 16;                  \
                HIDX = (cnt + 1) * n + (lo - kClRightLane) * V;                                                      \
                have = false;                                                                                        \
            }                                                                                                        \
            /* the halo entries this lane is going to update: read now, under the wait */                           \
            if (HIDX != 0xffffffffu) {                                                                               \
                _Pragma("unroll") for (int j_ = 0; j_ < V; ++j_) HV[j_] = (HWIN)[HIDX + j_];                         \
            }                                                                                                        \
            cl_u32x4 got = {0u, 0u, 0u, 0u};                                                                         \
            OK = true;                                                                                               \
            if constexpr (ONE) {                                                                                     \
                got = lpart[lo < PPM ? lo : 0u];                                                                     \
                OK = blk != drop_block;                                                                              \
            } else                                                                                                   \
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
            GOTX = got.x;                                                                                            \
            GOTZ = got.z;                                                                                            \
            cl_unpack<T, V>(GOTX, GOTZ, GV);                                                                         \
            TOTAL = wave_sum(lo < PPM * H ? cl_partials<T>(GOTX, GOTZ) : T(0));                                      \
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
    if (tid == 0) reinterpret_cast<u64 *>(ws + kClCtrlBytes + 2u * 512u * kClSlotBytes)[2 * blk + (WHICH)] = __builtin_amdgcn_s_memrealtime();
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
            if (lo == 0) {
                if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b64(z2, region, (int)(o + wave * 8), 0, kClSc1);
                else __builtin_amdgcn_raw_buffer_store_b128(z4, region, (int)(o + wave * 16), 0, kClSc1);
            }
            if (wave == 0 && has_left && lo < LPB) __builtin_amdgcn_raw_buffer_store_b128(z4, region, (int)(o + kClFirstOff + lo * 16), 0, kClSc1);
            if (wave == wl && has_right && lo - lb * LPB < LPB)
                __builtin_amdgcn_raw_buffer_store_b128(z4, region, (int)(o + kClLastOff + (lo - lb * LPB) * 16), 0, kClSc1);
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
        T total = T(0), gv[V], hv[V], part;
#pragma unroll
        for (int j = 0; j < V; ++j) gv[j] = hv[j] = T(0);
        uint32_t hidx = 0xffffffffu, gotx = 0u, gotz = 0u;
        bool ok = true;
        if (!greeted) {
            // HELLO (epoch 1): where does everybody run?  The payload of every partial granule is the sender's XCD id.
            greeted = true;
            T my_id, zeros[V];   // a partial whose words both read xcc_id
#pragma unroll
            for (int j = 0; j < V; ++j) zeros[j] = T(0);
            if constexpr (sizeof(T) == 4) my_id = __builtin_bit_cast(float, xcc_id);
            else my_id = __builtin_bit_cast(double, ((u64)xcc_id << 32) | xcc_id);
            if (tid == 0) bci[0] = 0u;
            wg_barrier();
            GBDPCG_CL_HANDOFF(1u, my_id, zeros, xa, total, gv, hv, hidx, ok, gotx, gotz)
            if (wave == POLL) {
                uint32_t lo = lane;
                asm volatile("" : "+v"(lo));
                const bool mine = lo >= PPM * H || (gotx == xcc_id && gotz == xcc_id);
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
        const T *S = a.S + prob * mstride;
        const T *P = a.Pinv ? a.Pinv + prob * mstride : nullptr;   // nullptr: identity preconditioner
        const size_t voff = (size_t)prob * len;

        // lambda and gamma of this lane's rows and the two halo knots of lambda (straight from the input vector: no member
        // writes lambda before every member has passed its first hand-off) are requested next to the first tile stages, so
        // that their round trips run under the 300 KB of tile loads instead of in front of the first product
        T lamv[V], gamv[V], halo_lam = T(0);
        auto request_vectors = [&]() {
            uint32_t lo = lane;
            asm volatile("" : "+v"(lo));   // addresses from the lane number, not from registers held since the kernel started
            const size_t g0 = voff + (size_t)k_lo * n + wave * (Dg::BPW * n) + V * lo;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                lamv[j] = dc.live ? a.lambda[g0 + j] : T(0);
                gamv[j] = dc.live ? a.gamma[g0 + j] : T(0);
            }
            const uint32_t t = wave * 64 + lo;
            if (t < 2 * n) {   // threads [0, n): the knot before the own ones, [n, 2n): the knot after them
                const int64_t gi = t < n ? (int64_t)k_lo * n - n + t : (int64_t)(k_lo + cnt) * n + (t - n);
                if (gi >= 0 && gi < (int64_t)len) halo_lam = a.lambda[voff + gi];
            }
        };

        DenseTile<T, NCT, V> tS, tP;
        if constexpr (STAGED) {
            GBDPCG_CL_STAMP(12, 0, ordinal == 0)
            dense_staged_load<T, NCT, V, PTAIL>(S, P, N, dc, wave, lane, k_lo, cnt, stage_raw, tS, tP,
                                      request_vectors);   // the vectors are requested behind the first two tile stages
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last picks are in registers before anything else happens
            GBDPCG_CL_STAMP(13, 0, ordinal == 0)
            GBDPCG_CL_STAMP_RT(15, 0, ordinal == 3)
        } else {
            request_vectors();
            dense_load<T, NCT, V>(S, N, dc, tS);
            if (P) {
                // (direct loads: the tail goes through the registers, once per problem -- and first, so that its registers are free
                // again when the columns that stay arrive)
                dense_load<T, NCT, V, Dg::COLS - PTAIL, Dg::COLS>(P, N, dc, tP);
                if constexpr (PTAIL > 0) {
#pragma unroll
                    for (int t = 0; t < PTAIL; ++t) {
                        if constexpr (V == 2) ptail[t * 64 + lane] = TV{tP.a[Dg::COLS - PTAIL + t][0], tP.a[Dg::COLS - PTAIL + t][1]};
                        else ptail[t * 64 + lane] = tP.a[Dg::COLS - PTAIL + t][0];
                    }
                    asm volatile("" ::: "memory");
                }
                dense_load<T, NCT, V, 0, Dg::COLS - PTAIL>(P, N, dc, tP);
            } else {
#pragma unroll
                for (uint32_t cc = 0; cc < Dg::COLS; ++cc)
#pragma unroll
                    for (int j = 0; j < V; ++j) tP.a[cc][j] = T(0);
            }
        }

        T rv[V], pv[V], yv[V];
        // windows: lambda on the own knots and the halos, zeros behind them (rows past the own knots, halos at the ends
        // of the problem); every entry has exactly one writer
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xa[n + row0 + j] = lamv[j];
        }
        if (tid < 2 * n) {
            const uint32_t i = tid < n ? tid : (cnt + 1) * n + (tid - n);
            xa[i] = halo_lam;
            xb[i] = T(0);
        }
        {
            uint32_t t = tid;
            asm volatile("" : "+v"(t));
            for (uint32_t i = (cnt + 2) * n + t; i < WINF; i += THREADS) xa[i] = xb[i] = T(0);
        }
        if (tid == 0) bci[0] = 0u;
        wg_barrier();
        GBDPCG_CL_STAMP_RT(16, 0, ordinal == 3)

        // r = gamma - S lambda                                            (pcg.cuh:118-126); the boundary knots of r travel
        dense_mv<T, NCT, V, CHAINS>(tS, xa, dc, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) rv[j] = dc.live ? gamv[j] - yv[j] : T(0);
        if (dc.live) {
#pragma unroll
            for (int j = 0; j < V; ++j) xb[n + row0 + j] = rv[j];
        }
        GBDPCG_CL_HANDOFF(e0 + 1u, T(0), rv, xb, total, gv, hv, hidx, ok, gotx, gotz)
        if (wave == POLL) {
            if (!ok && lane == 0) bci[0] = 2u;
            if (ok && hidx != 0xffffffffu) {
#pragma unroll
                for (int j = 0; j < V; ++j) xb[hidx + j] = gv[j];
            }
        }
        wg_barrier();
        bool failed = bci[0] == 2u;
        GBDPCG_CL_STAMP_RT(17, 0, ordinal == 3)

        // r~ = Pinv r ; p = r~ ; eta = r.r~                               (pcg.cuh:130-149)
        T eta = T(0);
        if (!failed) {
            if (P) dense_mv<T, NCT, V, CHAINS, PTAIL>(tP, xb, dc, yv, pt_lane, 64, ptd_lane);
            part = T(0);
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
            GBDPCG_CL_HANDOFF(e0 + 2u, part, pv, xa, total, gv, hv, hidx, ok, gotx, gotz)
            if (wave == POLL) {
                if (lane == 0) {
                    if (!ok) bci[0] = 2u;
                    bc[0] = total;
                }
                if (ok && hidx != 0xffffffffu) {
#pragma unroll
                    for (int j = 0; j < V; ++j) xa[hidx + j] = gv[j];
                }
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
            dense_mv<T, NCT, V, CHAINS>(tS, xa, dc, yv);
            part = T(0);
#pragma unroll
            for (int j = 0; j < V; ++j) part = fma_t(pv[j], yv[j], part);
            part = wave_sum(part);
            GBDPCG_CL_STAMP(2, POLL, iter == 3 && ordinal == 0)
            GBDPCG_CL_STAMP(9, 0, iter == 3 && ordinal == 0)
            GBDPCG_CL_HANDOFF(e0 + 3u + 2u * iter, part, yv, xb, total, gv, hv, hidx, ok, gotx, gotz)
            GBDPCG_CL_STAMP(3, POLL, iter == 3 && ordinal == 0)
            GBDPCG_CL_STAMP(10, 0, iter == 3 && ordinal == 0)
            if (wave == POLL) {
                const T al = eta / total;
                if (lane == 0) {
                    if (!ok) bci[0] = 2u;
                    bc[0] = al;
                }
                // r -= alpha upsilon on the halo knots: the owner's fma on the owner's bits
                if (ok && hidx != 0xffffffffu) {
#pragma unroll
                    for (int j = 0; j < V; ++j) xb[hidx + j] = fma_t(-al, gv[j], hv[j]);
                }
            }
            wg_barrier();
            GBDPCG_CL_STAMP(4, POLL, iter == 3 && ordinal == 0)
            if (bci[0] == 2u) { failed = true; break; }
            const T alpha = bc[0];
            // lambda += alpha p ; r -= alpha upsilon                      (pcg.cuh:172-176)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                lamv[j] = fma_t(alpha, pv[j], lamv[j]);
                rv[j] = fma_t(-alpha, yv[j], rv[j]);
            }
            if (dc.live) {
                const uint32_t oi = own_idx();
#pragma unroll
                for (int j = 0; j < V; ++j) xb[oi + j] = rv[j];
            }
            wg_barrier();
            GBDPCG_CL_STAMP(5, POLL, iter == 3 && ordinal == 0)
            // r~ = Pinv r ; eta_new = r.r~                                (pcg.cuh:180-193)
            if (P) dense_mv<T, NCT, V, CHAINS, PTAIL>(tP, xb, dc, yv, pt_lane, 64, ptd_lane);
            part = T(0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (!P) yv[j] = rv[j];
                part = fma_t(rv[j], yv[j], part);
            }
            part = wave_sum(part);
            GBDPCG_CL_HANDOFF(e0 + 4u + 2u * iter, part, yv, xa, total, gv, hv, hidx, ok, gotx, gotz)
            if (wave == POLL) {
                const bool conv = fabs(total) < a.tol;                   // pcg.cuh:195
                const T be = total / eta;                             // pcg.cuh:199
                if (lane == 0) {
                    bci[0] = !ok ? 2u : (conv ? 1u : 0u);
                    bc[0] = total;
                    bc[1] = be;
                }
                // p = r~ + beta p on the halo knots                       (pcg.cuh:203-206)
                if (ok && !conv && hidx != 0xffffffffu) {
#pragma unroll
                    for (int j = 0; j < V; ++j) xa[hidx + j] = fma_t(be, hv[j], gv[j]);
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
            const T beta = bc[1];
            eta = bc[0];
#pragma unroll
            for (int j = 0; j < V; ++j) pv[j] = fma_t(beta, pv[j], yv[j]);
            if (dc.live) {
                const uint32_t oi = own_idx();
#pragma unroll
                for (int j = 0; j < V; ++j) xa[oi + j] = pv[j];
            }
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
            const size_t g = voff + (size_t)k_lo * n + wave * (Dg::BPW * n) + V * lo;
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
            T *vec = reinterpret_cast<T *>(a.rescue_vec) + (size_t)c * rescue_vec_elems<T>(n, N);
            for (uint32_t prob = from; prob < a.batch; prob += clusters)
                if (pcg_takes(a, prob)) stream_rescue<T, Dg::WAVES>(a, prob, vec, rescue_red);
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

// What the kernel is built for: X(element type, block size, rows per lane).  A lane's rows of one column are 8 bytes at most
// (two fp32 rows, or one fp64 / fp32 row), n / V <= 16 lanes make a knot, and 2 x 3n x V x sizeof(T) / 4 matrix registers per
// lane must leave room for the working set: at n = 16 (fp32, V = 2: 192) the R block of Pinv stays in LDS (ClusterTail).
// fp64 at the BASELINE block size runs one row per lane: 14 lanes per knot, 32 knots per workgroup, four workgroups for N = 128.
// stateSize 13 (odd: one fp32 row per lane, direct tile loads -- its blocks are not whole 16-byte pieces -- and one accumulator
// chain per row).
#define GBDPCG_CLUSTER_SHAPES(X) \
    X(float, 2, 2) X(float, 3, 1) X(float, 4, 2) X(float, 5, 1) X(float, 6, 2) X(float, 7, 1) \
    X(float, 8, 2) X(float, 9, 1) X(float, 10, 2) X(float, 11, 1) X(float, 12, 2) X(float, 13, 1) X(float, 14, 2) X(float, 15, 1) X(float, 16, 2) \
    X(float, 18, 2) \
    X(double, 2, 1) X(double, 3, 1) X(double, 4, 1) X(double, 5, 1) X(double, 6, 1) X(double, 7, 1) X(double, 8, 1) X(double, 9, 1) X(double, 10, 1) X(double, 11, 1) X(double, 12, 1) \
    X(double, 13, 1) X(double, 14, 1) X(double, 15, 1) X(double, 16, 1)

// General storage, horizons beyond what ONE workgroup keeps in registers (pcg_resident.hip: 8 waves x floor(64 / (n / V)) knots
// -- 72 at n = 14 in fp32) up to eight (fp64: four) times that.  GBDPCG_NO_CLUSTER disables the path (tuning runs).
template <typename T> uint32_t cluster_members(uint32_t n, uint32_t N)
{
    static const bool off = getenv("GBDPCG_NO_CLUSTER") != nullptr;
    if (off) return 0;
    uint32_t per_wg = 0;
#define GBDPCG_X(TT, NN, VV) \
    if (sizeof(T) == sizeof(TT) && n == NN) per_wg = DenseGeom<TT, NN, VV>::MAX_KNOTS;
    GBDPCG_CLUSTER_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    if (per_wg == 0) return 0;                  // not built for the block size
    // one workgroup holds the whole problem: pcg_resident.hip where it is built for the block size (its reductions stay in LDS); a
    // "cluster" of one elsewhere (16 and 18, fp64 from 14 on) -- the kernel's ONE instantiation, whose wave partials meet in LDS
    if (N <= per_wg && resident_shape<T>(n, N)) return 0;
    static const bool no_single = getenv("GBDPCG_NO_CLUSTER_OF_ONE") != nullptr;   // tuning runs
    // (n = 2 below 16 knots stays with the streaming kernel, as in pcg_resident.hip: the reference's own example system lives there)
    if (N <= per_wg && (no_single || N < 2 || (n == 2 && N < 16))) return 0;
    const uint32_t H = (N + per_wg - 1) / per_wg;
    return H <= cl_max_members<T>() ? H : 0;
}

// ... plus 16 bytes per CU behind the slots: start / end of every workgroup on the real-time clock (diagnostic build only)
// (slots for two workgroups per CU: the light instantiations run two -- cluster_wgs_per_cu)
size_t cluster_workspace_bytes(const DeviceInfo &dev) { return kClCtrlBytes + (size_t)2 * (2 * dev.num_cus) * kClSlotBytes + (size_t)(2 * dev.num_cus) * 16; }

// Device memory for the in-kernel rescue: one set of vectors per cluster, sized for the longest horizon the path takes.
size_t cluster_rescue_bytes(const DeviceInfo &dev)
{
    size_t bytes = 0;
#define GBDPCG_X(TT, NN, VV)                                                                                      \
    {                                                                                                              \
        const size_t e = rescue_vec_elems<TT>(NN, cl_max_members<TT>() * DenseGeom<TT, NN, VV>::MAX_KNOTS) * sizeof(TT); \
        bytes = e > bytes ? e : bytes;                                                                             \
    }
    GBDPCG_CLUSTER_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    return (size_t)dev.num_cus * bytes;   // (a cluster may be a single workgroup)
}

template <typename T>
bool launch_pcg_cluster(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err)
{
    const uint32_t H = cluster_members<T>(a.n, a.N);
    if (H == 0 || !a.cluster_ws || a.symmetric) return false;
    if ((reinterpret_cast<uintptr_t>(a.S) % 8) || (a.Pinv && reinterpret_cast<uintptr_t>(a.Pinv) % 8)) return false;
    // one workgroup per CU: a cluster's members are resident together -- two where the instantiation is light enough for two to
    // share a CU (ClusterLight: at most 128 VGPRs, no dynamic LDS, and always two or more members, pcg_resident.hip having the
    // short horizons: so never more than CU-count clusters, which is what the control words hold)
    static const int per_cu_env = [] { const char *e = getenv("GBDPCG_CLUSTER_PER_CU"); return e ? atoi(e) : 0; }();   // tuning runs
    bool light = false;
#define GBDPCG_X(TT, NN, VV) \
    if (sizeof(T) == sizeof(TT) && a.n == NN) light = ClusterLight<TT, NN>::value && H >= 2;
    GBDPCG_CLUSTER_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    const uint32_t per_cu = light ? (per_cu_env ? (uint32_t)per_cu_env : 2u) : 1u;
    uint32_t clusters = (uint32_t)dev.num_cus * (per_cu > 2 ? 2u : per_cu) / H;
    if (clusters > a.batch) clusters = a.batch;
    if (clusters == 0) return false;
    const uint32_t rounds = (a.batch + clusters - 1) / clusters;
    // a tag = {launch number, epoch}: the epochs of a launch take the low kClEpochBits of its 32 bits
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
    const bool aligned16 = !((reinterpret_cast<uintptr_t>(a.S) | reinterpret_cast<uintptr_t>(a.Pinv)) % 16);
    static const bool no_plain = getenv("GBDPCG_CLUSTER_NO_PLAIN") != nullptr;   // tuning runs: always sc1 stores
    bool launched = false;
#define GBDPCG_X(TT, NN, VV)                                                                                             \
    if constexpr (sizeof(T) == sizeof(TT)) {                                                                             \
        if (a.n == NN && !launched) {                                                                                    \
            constexpr bool can_stage = DenseStage<TT, NN, VV>::OK;                                                       \
            const bool staged = can_stage && !no_staging && aligned16;                                                   \
            const size_t tail = (size_t)ClusterTail<TT, NN>::COLS * 512 * 8;                                \
            size_t lds = tail;                                                                                           \
            void (*kern)(PcgArgs<T>, unsigned char *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, bool) =          \
                pcg_cluster_kernel<T, NN, VV, false>;                                                                    \
            if constexpr (can_stage) {                                                                                   \
                if (staged) {                                                                                            \
                    kern = pcg_cluster_kernel<T, NN, VV, true>;                                                          \
                    if (dense_stage_lds_bytes<TT, NN, VV>() > lds) lds = dense_stage_lds_bytes<TT, NN, VV>();            \
                }                                                                                                        \
            }                                                                                                            \
            /* a cluster of one: the instantiation whose partials meet in LDS, where the block size has such launches   \
               (no single-workgroup kernel in pcg_resident.hip: fp32 16 and 18, fp64 14, 15, 16) */                      \
            if constexpr (!ClusterLight<TT, NN>::value && ((sizeof(TT) == 4 && NN >= 16) || (sizeof(TT) == 8 && NN >= 14))) { \
                if (H == 1) {                                                                                            \
                    kern = pcg_cluster_kernel<T, NN, VV, false, true>;                                                   \
                    if constexpr (can_stage) {                                                                           \
                        if (staged) kern = pcg_cluster_kernel<T, NN, VV, true, true>;                                    \
                    }                                                                                                    \
                }                                                                                                        \
            }                                                                                                            \
            /* on every launch, like the other launchers: HIP keeps the attribute per device */                          \
            if (lds) {                                                                                                   \
                *err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (*err != hipSuccess) return true;                                                                     \
            }                                                                                                            \
            hipLaunchKernelGGL(kern, dim3(clusters * H), dim3(512), lds, s, ka, static_cast<unsigned char *>(a.cluster_ws), H, C, \
                               clusters, spin_limit, drop_block, no_plain);                                              \
            launched = true;                                                                                             \
        }                                                                                                                \
    }
    GBDPCG_CLUSTER_SHAPES(GBDPCG_X)
#undef GBDPCG_X
    if (!launched) return false;
    *err = hipGetLastError();
    return true;
}

template uint32_t cluster_members<float>(uint32_t, uint32_t);
template uint32_t cluster_members<double>(uint32_t, uint32_t);
template bool launch_pcg_cluster<float>(const DeviceInfo &, const PcgArgs<float> &, hipStream_t, hipError_t *);
template bool launch_pcg_cluster<double>(const DeviceInfo &, const PcgArgs<double> &, hipStream_t, hipError_t *);

}  // namespace gbdpcg
