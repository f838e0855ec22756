// hop_probe.hip -- how long does one hand-off between two workgroups take, by store flavour and by placement?
//
// 256 workgroups (one per CU: the dynamic LDS request keeps a second one out), paired either (b, b + 8) -- the same XCD
// under round-robin dispatch -- or (b, b + 1) -- neighbouring XCDs.  Each pair plays ping-pong with data-tagged 8-byte
// granules {epoch, value} (MI355X_MICROARCH.md, Valid forms, R2): the producer stores, the consumer polls with sc1
// loads (served by L2), answers, and so on for ROUNDS round trips; all pairs play at the same time, as the workgroups of
// a clustered solve would.  Store flavours: sc1 (write-through: valid for any placement), plain and sc0 (the line stays in
// the XCD's L2: only a same-XCD poller is entitled to see it -- pairs whose partner never sees the word time out after a
// short bounded spin instead of hanging).  Prints per variant: median / max round-trip time over the pairs, pairs on one XCD.
//   hipcc -O2 --offload-arch=gfx950 hop_probe.hip -o hop_probe && ./hop_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kSc1 = 16;

struct Result {
    u64 ticks;       // 100 MHz real-time ticks for ROUNDS round trips (stamped by member 0 of the pair)
    uint32_t xcc[2]; // XCC id of the two members
    uint32_t ok;     // 1: all rounds completed, 0: timed out
    uint32_t bad;    // payload mismatches
};

template <int STORE_AUX>
__global__ __launch_bounds__(64) void hop_kernel(u64 *slots, Result *res, uint32_t rounds, uint32_t stride_pairs, uint32_t spin_limit,
                                                  uint32_t shift)
{
    extern __shared__ unsigned char lds_hog[];
    (void)lds_hog;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    // pairing: stride_pairs = 8 -> (b, b + 8); 1 -> (b, b + 1)
    const uint32_t grp = b / (2 * stride_pairs), in = b % (2 * stride_pairs);
    const uint32_t member = in / stride_pairs, pair = grp * stride_pairs + in % stride_pairs;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (lane == 0) res[pair].xcc[member] = xcc;
    // each member owns one 128-byte line; it writes its own, polls the partner's
    // a block's line is the one of block (b + shift): shift != 0 moves every line to a writer on the NEXT XCD, while the
    // poller of a line can be the block that wrote it (and may still hold it in its L2) in an earlier launch
    const uint32_t partner = member == 0 ? b + stride_pairs : b - stride_pairs;
    u64 *mine = slots + (size_t)((b + shift) % gridDim.x) * 16, *theirs = slots + (size_t)((partner + shift) % gridDim.x) * 16;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, 128, 0x00020000);
    const uint32_t base = (uint32_t)(*mine >> 32);   // epochs continue from what the last launch left
    uint32_t bad = 0;
    bool ok = true;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        for (uint32_t r = 1; r <= rounds && ok; ++r) {
            const uint32_t epoch = base + r;
            if (member == 0) {   // ping
                const u32x2 x = {r * 7u, epoch};
                __builtin_amdgcn_raw_buffer_store_b64(x, rm, 0, 0, STORE_AUX);
            }
            // wait for the partner's word of this epoch
            uint32_t spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");   // the poll must be re-issued every pass
                const u32x2 y = __builtin_amdgcn_raw_buffer_load_b64(rt, 0, 0, kSc1);
                if (y.y == epoch) {
                    if (y.x != (member == 0 ? r * 7u + 1u : r * 7u)) ++bad;
                    break;
                }
                if (++spins >= spin_limit) { ok = false; break; }
            }
            if (member == 1 && ok) {   // pong
                const u32x2 x = {r * 7u + 1u, epoch};
                __builtin_amdgcn_raw_buffer_store_b64(x, rm, 0, 0, STORE_AUX);
            }
        }
        const u64 t1 = __builtin_amdgcn_s_memrealtime();
        if (member == 0) {
            res[pair].ticks = t1 - t0;
            res[pair].ok = ok ? 1u : 0u;
            res[pair].bad = bad;
        }
        // leave the epoch base of the next launch in BOTH lines' tags (a timed-out pair skips ahead too)
        if (!ok) {
            const u32x2 x = {0u, base + rounds};
            __builtin_amdgcn_raw_buffer_store_b64(x, rm, 0, 0, kSc1);
        }
    }
}

template <int AUX>
static void run(const char *tag, u64 *slots, Result *res, uint32_t wgs, uint32_t rounds, uint32_t stride_pairs, uint32_t shift = 0)
{
    const uint32_t pairs = wgs / 2;
    CK(hipMemset(res, 0, sizeof(Result) * pairs));
    auto kern = hop_kernel<AUX>;
    const int lds = 100 * 1024;   // one workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 3; ++rep) {   // the last repetition is reported
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), lds, nullptr, slots, res, rounds, stride_pairs, AUX == 16 ? 1u << 20 : 1u << 14, shift);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    std::vector<Result> h(pairs);
    CK(hipMemcpy(h.data(), res, sizeof(Result) * pairs, hipMemcpyDeviceToHost));
    std::vector<double> same, cross;
    uint32_t timed_out = 0, bad = 0;
    for (auto &r : h) {
        bad += r.bad;
        if (!r.ok) { ++timed_out; continue; }
        const double us = (double)r.ticks / 100.0 / rounds;   // 100 MHz ticks -> us per round trip
        (r.xcc[0] == r.xcc[1] ? same : cross).push_back(us);
    }
    auto stat = [](std::vector<double> &v, double &med, double &mx) {
        if (v.empty()) { med = mx = 0; return; }
        std::sort(v.begin(), v.end());
        med = v[v.size() / 2];
        mx = v.back();
    };
    double ms, xs, mc, xc;
    stat(same, ms, xs);
    stat(cross, mc, xc);
    printf("%-44s pairs on one XCD %3zu: round trip median %.3f us max %.3f | across XCDs %3zu: median %.3f max %.3f | timed out %u, bad payloads %u\n",
           tag, same.size(), ms, xs, cross.size(), mc, xc, timed_out, bad);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t wgs = (uint32_t)prop.multiProcessorCount & ~15u;
    u64 *slots;
    Result *res;
    CK(hipMalloc((void **)&slots, (size_t)wgs * 128));
    CK(hipMemset(slots, 0, (size_t)wgs * 128));
    CK(hipMalloc((void **)&res, sizeof(Result) * wgs));
    printf("%u workgroups (one per CU), 200 round trips per pair, all pairs at once; one round trip = two hand-offs\n", wgs);
    run<16>("sc1 stores, pairs (b, b+8)", slots, res, wgs, 200, 8);
    run<16>("sc1 stores, pairs (b, b+1)", slots, res, wgs, 200, 1);
    run<0>("plain stores (no cache bits), pairs (b, b+8)", slots, res, wgs, 200, 8);
    run<0>("plain stores (no cache bits), pairs (b, b+1)", slots, res, wgs, 200, 1);
    run<1>("sc0 stores, pairs (b, b+8)", slots, res, wgs, 200, 8);
    run<1>("sc0 stores, pairs (b, b+1)", slots, res, wgs, 200, 1);
    run<16>("sc1 stores, pairs (b, b+8), again", slots, res, wgs, 200, 8);
    // Does a line that an XCD wrote with plain stores in one launch (it stays in that L2) go stale for that XCD's pollers
    // when another XCD writes it in a later launch?  plain (b, b+8), then sc1 (b, b+1) with every line moved one block on:
    // block b+1 now polls the line it wrote itself in the launch before, which block b (the previous XCD) writes now.
    run<0>("plain stores, pairs (b, b+8), lines in place", slots, res, wgs, 200, 8);
    run<16>("sc1 stores, pairs (b, b+1), lines moved by one block", slots, res, wgs, 200, 1, 1);
    run<0>("plain stores, pairs (b, b+8), lines in place", slots, res, wgs, 200, 8);
    run<16>("sc1 stores, pairs (b, b+8), lines moved by one block", slots, res, wgs, 200, 8, 1);
    return 0;
}
