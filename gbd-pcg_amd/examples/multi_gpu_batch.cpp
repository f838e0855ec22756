// Batched-problem mode over the GPUs of one node, in C++ against the C ABI (include/gbdpcg.h): BASELINE.json
// configs[4] -- stateSize 14, knotPoints 128, fp32, a batch of independent problems sharded over up to 8 MI355X.
// The reference has no multi-GPU code; this is the host driver the north star asks for:
//
//   * ONE process, one host thread per device; each thread owns a handle, a stream, its contiguous shard of the
//     batch (problems [g B/G, (g+1) B/G), built and kept on that device) and one hipGraph of
//     { Phi^-1 = symmetric stair from S ; solve }
//   * no collective touches problem data: the shards never leave their GPU
//   * RCCL over xGMI only for the throughput aggregation: ncclAllReduce(sum) of {problems solved, sum of iteration
//     counts} and ncclAllReduce(max) of the elapsed time, one communicator per device (ncclCommInitAll)
//
// usage: multi_gpu_batch [batch=8192] [knotPoints=128] [steps=5] [devices=all]
// Exit code 0 only if every device's solves converged with a small true residual and the aggregated problem count
// equals the batch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gbdpcg.h"
#include "synth_problem.hpp"

namespace {

struct Shared {
    uint32_t batch, N, devices;
    int steps;
    std::vector<ncclComm_t> comms;
    std::atomic<uint32_t> arrived{0}, finished{0};
    std::atomic<int> failures{0};
};

struct Report {
    uint64_t solved = 0, iter_sum = 0;       // job totals after the all-reduce
    double max_ms = 0.0;
    double own_ms = 0.0, worst_residual = 0.0;
    uint32_t lo = 0, hi = 0;
};

#define HIP_OK(x)                                                                                            \
    do {                                                                                                     \
        hipError_t e_ = (x);                                                                                 \
        if (e_ != hipSuccess) {                                                                              \
            fprintf(stderr, "[dev %d] HIP error %s at %s:%d\n", dev, hipGetErrorString(e_), __FILE__, __LINE__); \
            sh.failures++;                                                                                   \
            return;                                                                                          \
        }                                                                                                    \
    } while (0)
#define PCG_OK(x)                                                                                                 \
    do {                                                                                                          \
        gbdpcg_status s_ = (x);                                                                                   \
        if (s_ != GBDPCG_OK) {                                                                                    \
            fprintf(stderr, "[dev %d] gbdpcg error %s at %s:%d\n", dev, gbdpcg_status_string(s_), __FILE__, __LINE__); \
            sh.failures++;                                                                                        \
            return;                                                                                               \
        }                                                                                                         \
    } while (0)
#define NCCL_OK(x)                                                                                              \
    do {                                                                                                        \
        ncclResult_t r_ = (x);                                                                                  \
        if (r_ != ncclSuccess) {                                                                                \
            fprintf(stderr, "[dev %d] RCCL error %s at %s:%d\n", dev, ncclGetErrorString(r_), __FILE__, __LINE__); \
            sh.failures++;                                                                                      \
            return;                                                                                             \
        }                                                                                                       \
    } while (0)

void device_thread(Shared &sh, int dev, Report &rep)
{
    const uint32_t N = sh.N;
    const size_t msz = (size_t)3 * n * n * N, vsz = (size_t)n * N;
    // contiguous shard: sizes differ by at most one
    const uint32_t base = sh.batch / sh.devices, extra = sh.batch % sh.devices;
    const uint32_t lo = dev * base + std::min<uint32_t>(dev, extra), cnt = base + ((uint32_t)dev < extra ? 1 : 0);
    rep.lo = lo;
    rep.hi = lo + cnt;

    HIP_OK(hipSetDevice(dev));
    gbdpcg_handle_t h = nullptr;
    PCG_OK(gbdpcg_create(&h, dev));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // the shard is built for this device only; problem i depends on (1234 + i) alone.  A few distinct systems per
    // shard are generated on the host (O(n^4 N) each) and tiled, the right-hand sides scaled per problem.
    const uint32_t distinct = std::min<uint32_t>(cnt, 8);
    std::vector<float> hS(msz * cnt), hg(vsz * cnt);
    for (uint32_t b = 0; b < distinct; ++b) make_problem(N, 1234 + lo + b, hS.data() + b * msz, hg.data() + b * vsz);
    for (uint32_t b = distinct; b < cnt; ++b) {
        std::copy(hS.begin() + (b % distinct) * msz, hS.begin() + (b % distinct + 1) * msz, hS.begin() + b * msz);
        const float scale = 1.0f + 0.001f * (float)((lo + b) % 97);
        for (size_t i = 0; i < vsz; ++i) hg[b * vsz + i] = hg[(b % distinct) * vsz + i] * scale;
    }

    float *dS, *dP, *dg, *dl, *dy;
    uint32_t *d_iters;
    uint8_t *d_flags;
    unsigned long long *d_counts;   // {solved, iteration sum} for the all-reduce
    double *d_time;
    HIP_OK(hipMalloc((void **)&dS, msz * cnt * 4));
    HIP_OK(hipMalloc((void **)&dP, msz * cnt * 4));
    HIP_OK(hipMalloc((void **)&dg, vsz * cnt * 4));
    HIP_OK(hipMalloc((void **)&dl, vsz * cnt * 4));
    HIP_OK(hipMalloc((void **)&dy, vsz * cnt * 4));
    HIP_OK(hipMalloc((void **)&d_iters, cnt * 4));
    HIP_OK(hipMalloc((void **)&d_flags, cnt));
    HIP_OK(hipMalloc((void **)&d_counts, 16));
    HIP_OK(hipMalloc((void **)&d_time, 8));
    HIP_OK(hipMemcpy(dS, hS.data(), msz * cnt * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dg, hg.data(), vsz * cnt * 4, hipMemcpyHostToDevice));

    gbdpcg_graph_t graph = nullptr;
    PCG_OK(gbdpcg_graph_create_form_pinv_solve_f32(h, n, N, cnt, dS, dP, GBDPCG_PINV_STAIR, dg, dl, nullptr, nullptr, 1e-6f,
                                                   100, d_iters, d_flags, &graph));
    // warm-up replay, then all devices start their timed replays together
    HIP_OK(hipMemsetAsync(dl, 0, vsz * cnt * 4, stream));
    PCG_OK(gbdpcg_graph_launch(graph, stream));
    HIP_OK(hipStreamSynchronize(stream));
    sh.arrived++;
    while (sh.arrived.load() < sh.devices && sh.failures.load() == 0) std::this_thread::yield();

    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < sh.steps; ++s) {
        HIP_OK(hipMemsetAsync(dl, 0, vsz * cnt * 4, stream));   // a control step starts from lambda = 0
        PCG_OK(gbdpcg_graph_launch(graph, stream));
    }
    HIP_OK(hipStreamSynchronize(stream));
    rep.own_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    // results of the last step: iteration counts, flags, true residual of every problem
    std::vector<uint32_t> it(cnt);
    std::vector<uint8_t> fl(cnt);
    std::vector<float> hy(vsz * cnt);
    HIP_OK(hipMemcpy(it.data(), d_iters, cnt * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(fl.data(), d_flags, cnt, hipMemcpyDeviceToHost));
    PCG_OK(gbdpcg_spmv_f32(h, n, N, cnt, dS, dl, dy, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(hy.data(), dy, vsz * cnt * 4, hipMemcpyDeviceToHost));
    unsigned long long counts[2] = {0, 0};
    for (uint32_t b = 0; b < cnt; ++b) {
        double rr = 0, gg = 0;
        for (size_t i = 0; i < vsz; ++i) {
            const double d = (double)hg[b * vsz + i] - hy[b * vsz + i];
            rr += d * d;
            gg += (double)hg[b * vsz + i] * hg[b * vsz + i];
        }
        const double rel = std::sqrt(rr / gg);
        rep.worst_residual = std::max(rep.worst_residual, rel);
        if (!fl[b] && rel < 1e-2) counts[0] += 1;
        counts[1] += it[b];
    }

    // every device thread must reach the collective, or nobody enters it
    sh.finished++;
    while (sh.finished.load() < sh.devices && sh.failures.load() == 0) std::this_thread::yield();
    if (sh.failures.load()) return;

    // RCCL: aggregation of the throughput numbers only
    HIP_OK(hipMemcpy(d_counts, counts, 16, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_time, &rep.own_ms, 8, hipMemcpyHostToDevice));
    NCCL_OK(ncclAllReduce(d_counts, d_counts, 2, ncclUint64, ncclSum, sh.comms[dev], stream));
    NCCL_OK(ncclAllReduce(d_time, d_time, 1, ncclDouble, ncclMax, sh.comms[dev], stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(counts, d_counts, 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&rep.max_ms, d_time, 8, hipMemcpyDeviceToHost));
    rep.solved = counts[0];
    rep.iter_sum = counts[1];

    PCG_OK(gbdpcg_graph_destroy(graph));
    PCG_OK(gbdpcg_destroy(h));
    for (void *q : {(void *)dS, (void *)dP, (void *)dg, (void *)dl, (void *)dy, (void *)d_iters, (void *)d_flags,
                    (void *)d_counts, (void *)d_time})
        (void)hipFree(q);
    (void)hipStreamDestroy(stream);
}

}  // namespace

int main(int argc, char **argv)
{
    Shared sh;
    sh.batch = argc > 1 ? (uint32_t)atoi(argv[1]) : 8192;
    sh.N = argc > 2 ? (uint32_t)atoi(argv[2]) : 128;
    sh.steps = argc > 3 ? atoi(argv[3]) : 5;
    int present = 0;
    if (hipGetDeviceCount(&present) != hipSuccess || present < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    sh.devices = argc > 4 ? (uint32_t)atoi(argv[4]) : (uint32_t)present;
    if (sh.devices < 1 || sh.devices > (uint32_t)present || sh.batch < sh.devices || sh.N < 1) {
        fprintf(stderr, "usage: multi_gpu_batch [batch>=devices] [knotPoints] [steps] [devices<=%d]\n", present);
        return 2;
    }
    std::vector<int> ids(sh.devices);
    for (uint32_t g = 0; g < sh.devices; ++g) ids[g] = (int)g;
    sh.comms.resize(sh.devices);
    if (ncclCommInitAll(sh.comms.data(), (int)sh.devices, ids.data()) != ncclSuccess) {
        fprintf(stderr, "ncclCommInitAll failed\n");
        return 2;
    }
    int ranks_seen = 0;
    ncclCommCount(sh.comms[0], &ranks_seen);

    std::vector<Report> reps(sh.devices);
    std::vector<std::thread> threads;
    for (uint32_t g = 0; g < sh.devices; ++g) threads.emplace_back(device_thread, std::ref(sh), (int)g, std::ref(reps[g]));
    for (auto &t : threads) t.join();
    for (auto c : sh.comms) ncclCommDestroy(c);
    if (sh.failures.load()) return 1;

    for (uint32_t g = 0; g < sh.devices; ++g)
        printf("device %u: problems [%u, %u)  %.3f ms for %d steps  worst ||gamma - S lambda|| / ||gamma|| = %.2e\n", g,
               reps[g].lo, reps[g].hi, reps[g].own_ms, sh.steps, reps[g].worst_residual);
    const Report &r0 = reps[0];
    const double per_step_ms = r0.max_ms / sh.steps;
    printf("RCCL ranks = %d  problems solved = %llu of %u  iterations (last step) = %llu  max time = %.3f ms\n", ranks_seen,
           (unsigned long long)r0.solved, sh.batch, (unsigned long long)r0.iter_sum, r0.max_ms);
    printf("throughput = %.3e problem-iterations/s, %.3e solves/s over %u device(s) (Phi^-1 formation included)\n",
           (double)r0.iter_sum / (per_step_ms * 1e-3), (double)sh.batch / (per_step_ms * 1e-3), sh.devices);
    bool ok = r0.solved == sh.batch && ranks_seen == (int)sh.devices;
    for (const Report &r : reps) ok = ok && r.solved == r0.solved && r.iter_sum == r0.iter_sum;   // every rank got the totals
    return ok ? 0 : 1;
}
