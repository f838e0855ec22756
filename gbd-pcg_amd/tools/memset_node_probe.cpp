// Stand-alone probe (no libgbdpcg): do hipGraph MEMSET nodes misbehave when a captured graph is replayed back to back?
//
// Round 1 initialised the per-problem symmetry flags with hipMemsetAsync(flags, 1, batch, stream) inside the captured
// solve and saw garbage flags on back-to-back replays (every problem then took the slow general kernel); the calls were
// replaced by fill kernels.  hipMemsetAsync takes pointer, value and size BY VALUE, so there is no captured-argument
// lifetime to get wrong on the caller's side; this program reproduces the pattern with nothing but the HIP runtime to
// tell a runtime problem from a library one:
//
//   graph (stream capture, thread-local mode, non-blocking capture stream):
//       memset(flags, 1, n)  ->  audit kernel: count bytes != 1, then AND a pattern in (what the symmetry check did)
//   replayed R times back to back on (a) the legacy null stream, (b) a non-blocking stream, with and without an
//   eager kernel or an eager hipMemsetAsync (what lambda.zero_() issues) between replays; every replay's audit count is kept.
//
// Prints one line per scenario: replays in which the audit kernel saw a byte that the memset node should have set.
//   hipcc -O2 --offload-arch=gfx950 memset_node_probe.cpp -o memset_node_probe && ./memset_node_probe
#include <hip/hip_runtime.h>

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

__global__ void audit_kernel(unsigned char *flags, unsigned n, unsigned expect, unsigned *bad, unsigned *replay_counter)
{
    __shared__ unsigned wrong;
    if (threadIdx.x == 0) wrong = 0;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
        if (flags[i] != expect) atomicAdd(&wrong, 1u);
    __syncthreads();
    // what the consumer of the flags did in round 1: AND results in (leaves 0 / 1 mixtures behind for the next replay)
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) flags[i] &= (unsigned char)(i & 1u);
    // ~50 us of dependent work: several replays are then queued behind the one that runs
    float spin = (float)wrong;
    for (int i = 0; i < 20000; ++i) spin = __builtin_fmaf(spin, 1.0000001f, 1e-9f);
    if (threadIdx.x == 0) {
        const unsigned r = atomicAdd(replay_counter, 1u);
        bad[r] = wrong + (spin < -1.f ? 1u : 0u);
    }
}

__global__ void busy_kernel(float *x, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = 0.f;
}

static int scenario(const char *tag, hipStream_t launch_stream, int interleave, unsigned nbytes, int value, int replays)
{
    unsigned char *flags;
    unsigned *bad, *counter;
    float *scratch;
    const unsigned scratch_n = 1u << 20;
    CK(hipMalloc((void **)&flags, nbytes));
    CK(hipMalloc((void **)&bad, sizeof(unsigned) * (replays + 8)));
    CK(hipMalloc((void **)&counter, sizeof(unsigned)));
    CK(hipMalloc((void **)&scratch, scratch_n * sizeof(float)));
    CK(hipMemset(flags, 0xEE, nbytes));
    CK(hipMemset(bad, 0, sizeof(unsigned) * (replays + 8)));
    CK(hipMemset(counter, 0, sizeof(unsigned)));

    hipStream_t cs;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(flags, value, nbytes, cs));
    hipLaunchKernelGGL(audit_kernel, dim3(1), dim3(256), 0, cs, flags, nbytes, (unsigned)value, bad, counter);
    CK(hipStreamEndCapture(cs, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

    for (int r = 0; r < replays; ++r) {
        if (interleave == 1) hipLaunchKernelGGL(busy_kernel, dim3(scratch_n / 256), dim3(256), 0, launch_stream, scratch, scratch_n);
        if (interleave == 2) CK(hipMemsetAsync(scratch, 0, scratch_n * sizeof(float), launch_stream));   // what tensor.zero_() issues
        CK(hipGraphLaunch(exec, launch_stream));
    }
    CK(hipStreamSynchronize(launch_stream));
    CK(hipDeviceSynchronize());

    std::vector<unsigned> h(replays);
    unsigned done = 0;
    CK(hipMemcpy(h.data(), bad, sizeof(unsigned) * replays, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&done, counter, sizeof(unsigned), hipMemcpyDeviceToHost));
    int bad_replays = 0;
    unsigned worst = 0;
    for (int r = 0; r < replays; ++r) {
        if (h[r]) ++bad_replays;
        if (h[r] > worst) worst = h[r];
    }
    printf("%-58s bytes=%-6u value=%d replays=%d audited=%u  replays with wrong bytes: %d (worst %u of %u bytes)\n", tag, nbytes,
           value, replays, done, bad_replays, worst, nbytes);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    CK(hipStreamDestroy(cs));
    CK(hipFree(flags));
    CK(hipFree(bad));
    CK(hipFree(counter));
    CK(hipFree(scratch));
    return bad_replays;
}

int main()
{
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    int total = 0;
    for (unsigned nbytes : {1024u, 1028u, 64u, 8192u}) {
        for (int value : {1, 0}) {
            total += scenario("null stream, back to back", nullptr, 0, nbytes, value, 40);
            total += scenario("null stream, eager kernel between replays", nullptr, 1, nbytes, value, 40);
            total += scenario("null stream, eager hipMemsetAsync between replays", nullptr, 2, nbytes, value, 40);
            total += scenario("non-blocking stream, back to back", side, 0, nbytes, value, 40);
            total += scenario("non-blocking stream, eager kernel between replays", side, 1, nbytes, value, 40);
            total += scenario("non-blocking stream, eager hipMemsetAsync between replays", side, 2, nbytes, value, 40);
        }
    }
    printf(total ? "MEMSET NODES MISBEHAVED in %d replays (runtime: nothing of libgbdpcg is linked here)\n"
                 : "memset nodes behaved in every scenario (%d bad replays)\n",
           total);
    CK(hipStreamDestroy(side));
    return 0;
}
