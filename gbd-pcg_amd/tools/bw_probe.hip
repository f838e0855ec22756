// bw_probe.hip -- bandwidth calibration for the block-tridiagonal SpMV on MI355X.
// Measures, on the same device and in one process (interleaved rounds, medians):
//   read4   : pure streaming read, 16 B per lane (sum reduction)     -> the read ceiling
//   read2   : pure streaming read, 8 B per lane, 63 of 64 lanes live -> what the n=14 lane map can reach
//   spmv    : gbdpcg_spmv_f32 through the C ABI (the shipped kernel)
// Every launch reads a matrix that 3 x 308 MB of other traffic has pushed out of the Infinity Cache.
// Usage: bw_probe [batch=1024] [N=128] [reps=30]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/gbdpcg.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void read4_kernel(const float4* __restrict__ p, size_t n4, float* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n4; i += stride) { float4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) out[0] = acc;
}

// contiguous per-workgroup streaming, 16 B per lane: each WG owns a contiguous slab
__global__ __launch_bounds__(256) void read4_slab_kernel(const float4* __restrict__ p, size_t n4, float* out)
{
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = min(n4, lo + per);
    float acc = 0.f;
    size_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {
        float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < hi; i += 256) { float4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) out[0] = acc;
}

// 8 B per lane, 63 live lanes per wave, each wave streams contiguous 504-byte pieces (the n=14 map)
__global__ __launch_bounds__(256) void read2_kernel(const float2* __restrict__ p, size_t n2, float* out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t waves = (size_t)gridDim.x * 4;
    const size_t per = (n2 / 63 + waves - 1) / waves;           // pieces per wave
    size_t piece = ((size_t)blockIdx.x * 4 + wave) * per;
    const size_t end = min(n2 / 63, piece + per);
    float acc = 0.f;
    if (lane < 63) {
        for (; piece + 4 < end; piece += 5) {
            float2 a = p[piece * 63 + lane], b = p[(piece + 1) * 63 + lane], c = p[(piece + 2) * 63 + lane],
                   d = p[(piece + 3) * 63 + lane], e = p[(piece + 4) * 63 + lane];
            acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y + e.x + e.y;
        }
        for (; piece < end; ++piece) { float2 a = p[piece * 63 + lane]; acc += a.x + a.y; }
    }
    if (acc == 123.456f) out[0] = acc;
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// same as read4_slab / read2 with the non-temporal cache policy (global_load ... nt)
__global__ __launch_bounds__(256) void read4_slab_nt_kernel(const v4f* __restrict__ p, size_t n4, float* out)
{
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = min(n4, lo + per);
    float acc = 0.f;
    size_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {
        v4f a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 256),
            c = __builtin_nontemporal_load(p + i + 512), d = __builtin_nontemporal_load(p + i + 768);
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < hi; i += 256) { v4f a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <bool NT, bool SWEEP>
__global__ __launch_bounds__(256) void read2x_kernel(const v2f* __restrict__ p, size_t n2, float* out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t waves = (size_t)gridDim.x * 4;
    const size_t pieces = n2 / 63;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    float acc = 0.f;
    auto ld = [&](size_t piece) -> v2f {
        const v2f* q = p + piece * 63 + lane;
        if (NT) return __builtin_nontemporal_load(q);
        return *q;
    };
    if (lane < 63) {
        if (SWEEP) {
            // chip-wide in-order sweep: in round i the waves read 5*waves adjacent pieces
            for (size_t base = gw * 5; base + 4 < pieces; base += waves * 5) {
                v2f a = ld(base), b = ld(base + 1), c = ld(base + 2), d = ld(base + 3), e = ld(base + 4);
                acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y + e.x + e.y;
            }
        } else {
            const size_t per = (pieces + waves - 1) / waves;
            size_t piece = gw * per;
            const size_t end = min(pieces, piece + per);
            for (; piece + 4 < end; piece += 5) {
                v2f a = ld(piece), b = ld(piece + 1), c = ld(piece + 2), d = ld(piece + 3), e = ld(piece + 4);
                acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y + e.x + e.y;
            }
            for (; piece < end; ++piece) { v2f a = ld(piece); acc += a.x + a.y; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 128;
    const int reps = argc > 3 ? atoi(argv[3]) : 30;
    const uint32_t n = 14;
    const size_t melems = (size_t)3 * n * n * N * batch, velems = (size_t)n * N * batch;
    const size_t mbytes = melems * 4;
    // NB distinct matrices, visited round-robin: a buffer is re-read only after (NB-1)*mbytes of other
    // traffic, so with NB*mbytes >> 256 MiB every launch streams from HBM, not from the Infinity Cache
    const int NB = 4;
    float *M[NB], *x, *y, *out;
    {
        std::vector<float> h(melems);
        for (int b = 0; b < NB; ++b) {
            CK(hipMalloc(&M[b], mbytes));
            for (size_t i = 0; i < melems; ++i) h[i] = (float)(((i + b) * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
            CK(hipMemcpy(M[b], h.data(), mbytes, hipMemcpyHostToDevice));
        }
    }
    CK(hipMalloc(&x, velems * 4)); CK(hipMalloc(&y, velems * 4)); CK(hipMalloc(&out, 256));
    {
        std::vector<float> hx(velems);
        for (size_t i = 0; i < velems; ++i) hx[i] = (float)((i * 40503u) & 0xfff) / 4096.f - 0.5f;
        CK(hipMemcpy(x, hx.data(), velems * 4, hipMemcpyHostToDevice));
    }
    gbdpcg_handle_t h;
    if (gbdpcg_create(&h, 0) != GBDPCG_OK) { printf("gbdpcg_create failed\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s = nullptr;

    const int NV = 10;
    const char* names[NV] = {"read4 (grid-stride)", "read4 (slab/WG)", "read2 63-lane", "spmv C-ABI", "read4 2048 WG", "read4 slab nt", "read2 63-lane nt", "read2 sweep", "read2 sweep nt", "read2 sweep 4096WG"};
    std::vector<float> t[NV];
    int turn = 0;
    for (int r = 0; r < reps + 3; ++r) {
        for (int v = 0; v < NV; ++v) {
            const float* Mb = M[turn++ % NB];
            CK(hipEventRecord(e0, s));
            switch (v) {
            case 0: hipLaunchKernelGGL(read4_kernel, dim3(256 * 16), dim3(256), 0, s, (const float4*)Mb, melems / 4, out); break;
            case 1: hipLaunchKernelGGL(read4_slab_kernel, dim3(256 * 8), dim3(256), 0, s, (const float4*)Mb, melems / 4, out); break;
            case 2: hipLaunchKernelGGL(read2_kernel, dim3(256 * 8), dim3(256), 0, s, (const float2*)Mb, melems / 2, out); break;
            case 3: if (gbdpcg_spmv_f32(h, n, N, batch, Mb, x, y, s) != GBDPCG_OK) { printf("spmv failed\n"); return 1; } break;
            case 4: hipLaunchKernelGGL(read4_kernel, dim3(2048), dim3(256), 0, s, (const float4*)Mb, melems / 4, out); break;
            case 5: hipLaunchKernelGGL(read4_slab_nt_kernel, dim3(256 * 8), dim3(256), 0, s, (const v4f*)Mb, melems / 4, out); break;
            case 6: hipLaunchKernelGGL((read2x_kernel<true, false>), dim3(256 * 8), dim3(256), 0, s, (const v2f*)Mb, melems / 2, out); break;
            case 7: hipLaunchKernelGGL((read2x_kernel<false, true>), dim3(256 * 8), dim3(256), 0, s, (const v2f*)Mb, melems / 2, out); break;
            case 8: hipLaunchKernelGGL((read2x_kernel<true, true>), dim3(256 * 8), dim3(256), 0, s, (const v2f*)Mb, melems / 2, out); break;
            case 9: hipLaunchKernelGGL((read2x_kernel<false, true>), dim3(256 * 16), dim3(256), 0, s, (const v2f*)Mb, melems / 2, out); break;
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) t[v].push_back(ms);
        }
    }
    printf("matrix bytes per launch: %.1f MB (batch %u, N %u), %d matrices visited round-robin (cold reads)\n",
           mbytes / 1e6, batch, N, NB);
    for (int v = 0; v < NV; ++v) {
        const float ms = median(t[v]);
        const double bytes = v == 3 ? (double)batch * ((3.0 * N - 2) * n * n + 2.0 * n * N) * 4 : (double)mbytes;
        printf("%-22s median %.4f ms  min %.4f ms  -> %.0f GB/s (median)  %.0f GB/s (best)\n", names[v], ms,
               *std::min_element(t[v].begin(), t[v].end()), bytes / ms / 1e6,
               bytes / *std::min_element(t[v].begin(), t[v].end()) / 1e6);
    }
    gbdpcg_destroy(h);
    return 0;
}
