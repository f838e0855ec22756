// launch_probe.hip -- what one dependent kernel launch costs on this box (hipGraph replay and eager):
// the floor under the split PCG path, which needs two launches per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_flag(const unsigned* flag, unsigned* out) { if (flag[blockIdx.x & 3]) return; if (threadIdx.x == 9999) out[0] = 1; }
__global__ void k_lds(const unsigned* flag, unsigned* out) { extern __shared__ unsigned s[]; if (flag[0]) return; s[threadIdx.x] = 1; __syncthreads(); if (s[0] == 7) out[0] = 1; }

struct BigArgs { const unsigned* flag; unsigned* out; double* p[14]; unsigned a, b, c, d; };
// early exit first, then a large body (kept alive through `out`) so the code object is tens of KB
template <int ID> __global__ __launch_bounds__(256) void k_big(BigArgs g, int iter)
{
    if (g.flag[blockIdx.x & 3]) return;
    double acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = g.p[i % 14][threadIdx.x + i * 256 + ID];
#pragma unroll
    for (int r = 0; r < 40; ++r)
#pragma unroll
        for (int i = 0; i < 24; ++i) acc[i] = acc[i] * acc[(i + r + 1) % 24] + (double)(r + ID);
    double t = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) t += acc[i];
    if (t == 1.2345 + iter) g.out[0] = 1;
}

template <typename F> static float graph_pitch_us(F launch, int nk, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nk; ++i) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int r = 0; r < 12; ++r) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return t[t.size() / 2] * 1000.f / nk;
}

int main()
{
    unsigned *flag, *out; CK(hipMalloc(&flag, 64)); CK(hipMalloc(&out, 64)); CK(hipMemset(flag, 0, 64)); CK(hipMemset(out, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int wgs : {1, 64, 256, 1024}) {
        float a = graph_pitch_us([&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st); }, 200, s);
        float b = graph_pitch_us([&](hipStream_t st) { hipLaunchKernelGGL(k_flag, dim3(wgs), dim3(256), 0, st, flag, out); }, 200, s);
        float c = graph_pitch_us([&](hipStream_t st) { hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 4096, st, flag, out); }, 200, s);
        printf("%4d WGs x 256 thr: graph pitch per kernel  empty %.2f us   flag-load+exit %.2f us   lds+barrier %.2f us\n", wgs, a, b, c);
    }
    // flag = 1: every k_big exits at once, like the split-PCG launches after convergence
    CK(hipMemset(flag, 1, 64));
    BigArgs ba{}; ba.flag = flag; ba.out = out; for (auto& q : ba.p) q = (double*)out;
    for (int wgs : {64, 256}) {
        int it = 0;
        float same = graph_pitch_us([&](hipStream_t st) { hipLaunchKernelGGL(k_big<0>, dim3(wgs), dim3(256), 4096, st, ba, it); }, 200, s);
        int tog = 0;
        float alt = graph_pitch_us([&](hipStream_t st) {
            if (tog++ & 1) hipLaunchKernelGGL(k_big<1>, dim3(wgs), dim3(256), 4096, st, ba, it);
            else hipLaunchKernelGGL(k_big<2>, dim3(wgs), dim3(256), 4096, st, ba, it); }, 200, s);
        printf("%4d WGs: big kernel (early exit taken)  same kernel %.2f us   two kernels alternating %.2f us\n", wgs, same, alt);
    }
    return 0;
}
