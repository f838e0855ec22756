// sym_probe.hip -- prototype: block-tridiagonal SpMV that reads only [D_k | R_k] of a SYMMETRIC
// matrix (L_{k+1} = R_k^T) and forms the L contribution from R_k^T on the fly; n = 14, fp32.
// Measures it against gbdpcg_spmv_f32 (which reads all three blocks) on Infinity-Cache-cold data
// and checks it against a double-precision host product.  Decides whether the symmetric path is
// worth building into the library (2/3 of the bytes, more cross-lane work per row).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/gbdpcg.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int n = 14, NN = n * n;

template <int CTRL> __device__ __forceinline__ float dpp_add_f(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}
// sum over the 8 lanes of an aligned group; every lane of the group gets it
__device__ __forceinline__ float sum8(float v)
{
    v = dpp_add_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add_f<0x141>(v);  // row_half_mirror
    return v;
}
// sum over lanes with equal (lane & 7): the 8 groups of the wave
__device__ __forceinline__ float sum_groups(float v)
{
    v = dpp_add_f<0x128>(v);  // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmv_sym14(const float* __restrict__ M, const float* __restrict__ x,
                                                         float* __restrict__ y, int N)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                  // (N+1)*n : x plus one zero block
    float* yb = xs + (N + 1) * n;    // N*n     : D_k x_k + R_k x_{k+1}
    float* zb = yb + N * n;          // N*n     : R_{k-1}^T x_{k-1}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t prob = blockIdx.x;
    const int len = N * n;
    for (int i = tid; i < len; i += WAVES * 64) xs[i] = x[prob * len + i];
    for (int i = tid; i < n; i += WAVES * 64) { xs[len + i] = 0.f; zb[i] = 0.f; }

    const int g = lane >> 3, rp = lane & 7;
    const bool act = rp < 7;
    const float* Mp = M + prob * 3 * NN * (size_t)N + NN;  // skip L_0: row k's [D|R] starts at k*3n^2 + n^2
    int off[4];
    bool val[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = g + 8 * s;
        val[s] = act && c < 2 * n;
        off[s] = val[s] ? c * n + rp * 2 : 0;
    }
    constexpr int DEPTH = 4;
    float2 ring[DEPTH][4];
    const int rows = N / WAVES;  // prototype: N % WAVES == 0 and rows >= 2*DEPTH
#define ISSUE(q_, slot_)                                                             \
    do {                                                                             \
        const float* b_ = Mp + (size_t)(wave + WAVES * (q_)) * 3 * NN;               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) ring[slot_][s] = *reinterpret_cast<const float2*>(b_ + off[s]); \
    } while (0)
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) { ISSUE(j, j); __builtin_amdgcn_sched_barrier(0); }
    __syncthreads();

#define CONSUME(q_, slot_, refill_)                                                                     \
    do {                                                                                                \
        const int k = wave + WAVES * (q_);                                                              \
        const float* xk = xs + k * n;                                                                   \
        const float2 own = act ? *reinterpret_cast<const float2*>(xk + rp * 2) : make_float2(0.f, 0.f); \
        const bool lastrow = k == N - 1;                                                                \
        float a0 = 0.f, a1 = 0.f, t[3];                                                                 \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                 \
            const int c = g + 8 * s;                                                                    \
            float2 a = ring[slot_][s];                                                                  \
            const bool keep = val[s] && !(lastrow && c >= n);                                           \
            a.x = keep ? a.x : 0.f; a.y = keep ? a.y : 0.f;                                             \
            const float xv = xk[c < 2 * n ? c : 2 * n - 1];                                             \
            a0 = __builtin_fmaf(a.x, xv, a0); a1 = __builtin_fmaf(a.y, xv, a1);                         \
            if (s >= 1) t[s - 1] = c >= n ? __builtin_fmaf(a.y, own.y, a.x * own.x) : 0.f;              \
        }                                                                                               \
        if (refill_) ISSUE((q_) + DEPTH, slot_);                                                        \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) t[j] = sum8(t[j]);                                \
        a0 = sum_groups(a0); a1 = sum_groups(a1);                                                       \
        if (g == 0 && act) *reinterpret_cast<float2*>(yb + k * n + rp * 2) = make_float2(a0, a1);       \
        if (rp == 0 && !lastrow) {                                                                      \
            _Pragma("unroll") for (int s = 1; s < 4; ++s) {                                             \
                const int c = g + 8 * s;                                                                \
                if (c >= n && c < 2 * n) zb[(k + 1) * n + c - n] = t[s - 1];                            \
            }                                                                                           \
        }                                                                                               \
    } while (0)

    int q0 = 0;
    for (; q0 + 2 * DEPTH <= rows; q0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) CONSUME(q0 + j, j, true);
    }
    for (; q0 < rows; q0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (q0 + j < rows) CONSUME(q0 + j, j, q0 + j + DEPTH < rows);
    }
    __syncthreads();
    for (int i = tid; i < len; i += WAVES * 64) y[prob * len + i] = yb[i] + zb[i];
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pcg_like_sym14(const float* __restrict__ M0, const float* __restrict__ M1,
                                                             const float* __restrict__ x, float* __restrict__ y, int N,
                                                             int passes, int batch)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                  // (N+1)*n : x plus one zero block
    float* yb = xs + (N + 1) * n;    // N*n     : D_k x_k + R_k x_{k+1}
    float* zb = yb + N * n;          // N*n     : R_{k-1}^T x_{k-1}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (size_t prob = blockIdx.x; prob < (size_t)batch; prob += gridDim.x) {
    const int len = N * n;
    for (int i = tid; i < len; i += WAVES * 64) xs[i] = x[prob * len + i];
    for (int i = tid; i < n; i += WAVES * 64) { xs[len + i] = 0.f; zb[i] = 0.f; }

    const int g = lane >> 3, rp = lane & 7;
    const bool act = rp < 7;
    int off[4];
    bool val[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = g + 8 * s;
        val[s] = act && c < 2 * n;
        off[s] = val[s] ? c * n + rp * 2 : 0;
    }
    constexpr int DEPTH = 4;
    float2 ring[DEPTH][4];
    const int rows = N / WAVES;
    for (int pass = 0; pass < passes; ++pass) {
    const float* Mp = ((pass & 1) ? M1 : M0) + prob * 3 * NN * (size_t)N + NN;
#undef ISSUE
#define ISSUE(q_, slot_)                                                             \
    do {                                                                             \
        const float* b_ = Mp + (size_t)(wave + WAVES * (q_)) * 3 * NN;               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) ring[slot_][s] = *reinterpret_cast<const float2*>(b_ + off[s]); \
    } while (0)
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) { ISSUE(j, j); __builtin_amdgcn_sched_barrier(0); }
    __syncthreads();

#undef CONSUME
#define CONSUME(q_, slot_, refill_)                                                                     \
    do {                                                                                                \
        const int k = wave + WAVES * (q_);                                                              \
        const float* xk = xs + k * n;                                                                   \
        const float2 own = act ? *reinterpret_cast<const float2*>(xk + rp * 2) : make_float2(0.f, 0.f); \
        const bool lastrow = k == N - 1;                                                                \
        float a0 = 0.f, a1 = 0.f, t[3];                                                                 \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                 \
            const int c = g + 8 * s;                                                                    \
            float2 a = ring[slot_][s];                                                                  \
            const bool keep = val[s] && !(lastrow && c >= n);                                           \
            a.x = keep ? a.x : 0.f; a.y = keep ? a.y : 0.f;                                             \
            const float xv = xk[c < 2 * n ? c : 2 * n - 1];                                             \
            a0 = __builtin_fmaf(a.x, xv, a0); a1 = __builtin_fmaf(a.y, xv, a1);                         \
            if (s >= 1) t[s - 1] = c >= n ? __builtin_fmaf(a.y, own.y, a.x * own.x) : 0.f;              \
        }                                                                                               \
        if (refill_) ISSUE((q_) + DEPTH, slot_);                                                        \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) t[j] = sum8(t[j]);                                \
        a0 = sum_groups(a0); a1 = sum_groups(a1);                                                       \
        if (g == 0 && act) *reinterpret_cast<float2*>(yb + k * n + rp * 2) = make_float2(a0, a1);       \
        if (rp == 0 && !lastrow) {                                                                      \
            _Pragma("unroll") for (int s = 1; s < 4; ++s) {                                             \
                const int c = g + 8 * s;                                                                \
                if (c >= n && c < 2 * n) zb[(k + 1) * n + c - n] = t[s - 1];                            \
            }                                                                                           \
        }                                                                                               \
    } while (0)

    int q0 = 0;
    for (; q0 + 2 * DEPTH <= rows; q0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) CONSUME(q0 + j, j, true);
    }
    for (; q0 < rows; q0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            if (q0 + j < rows) CONSUME(q0 + j, j, q0 + j + DEPTH < rows);
    }
    __syncthreads();
    // feed the product back as the next pass's x (scaled to stay finite), like p / r in PCG
    for (int i = tid; i < len; i += WAVES * 64) xs[i] = 0.125f * (yb[i] + zb[i]);
    __syncthreads();
    }
    for (int i = tid; i < len; i += WAVES * 64) y[prob * len + i] = xs[i];
    __syncthreads();
    }
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 128, reps = argc > 3 ? atoi(argv[3]) : 20;
    const size_t melems = (size_t)3 * NN * N * batch, velems = (size_t)n * N * batch;
    const int NB = 4;
    float *M[NB], *x, *y, *y2;
    std::vector<float> h(melems);
    // symmetric block-tridiagonal: D_k symmetric, L_{k+1} = R_k^T exactly
    auto rnd = [](size_t i) { return (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f; };
    for (int b = 0; b < NB; ++b) {
        for (size_t p = 0; p < (size_t)batch; ++p)
            for (int k = 0; k < N; ++k) {
                float* blk = &h[(p * N + k) * 3 * NN];
                for (int c = 0; c < n; ++c)
                    for (int r = 0; r < n; ++r) {
                        const size_t id = ((p * N + k) * NN + std::min(r, c) * n + std::max(r, c)) * 7 + b;
                        blk[NN + c * n + r] = rnd(id) + (r == c ? 4.f : 0.f);         // D_k symmetric
                        blk[2 * NN + c * n + r] = rnd(id * 3 + 1 + (size_t)(r * n + c) * 1315423911u);  // R_k
                    }
            }
        for (size_t p = 0; p < (size_t)batch; ++p)
            for (int k = 0; k < N; ++k) {
                float* blk = &h[(p * N + k) * 3 * NN];
                for (int c = 0; c < n; ++c)
                    for (int r = 0; r < n; ++r)
                        blk[c * n + r] = k == 0 ? 1e30f : h[(p * N + k - 1) * 3 * NN + 2 * NN + r * n + c];  // L_k = R_{k-1}^T
            }
        CK(hipMalloc(&M[b], melems * 4));
        CK(hipMemcpy(M[b], h.data(), melems * 4, hipMemcpyHostToDevice));
    }
    std::vector<float> hx(velems);
    for (size_t i = 0; i < velems; ++i) hx[i] = rnd(i * 11 + 5);
    CK(hipMalloc(&x, velems * 4)); CK(hipMalloc(&y, velems * 4)); CK(hipMalloc(&y2, velems * 4));
    CK(hipMemcpy(x, hx.data(), velems * 4, hipMemcpyHostToDevice));

    gbdpcg_handle_t hd;
    if (gbdpcg_create(&hd, 0) != GBDPCG_OK) { printf("gbdpcg_create failed\n"); return 1; }
    const size_t lds = ((size_t)(N + 1) * n + 2 * (size_t)N * n) * 4;
    // correctness on the last matrix: symmetric kernel vs library kernel vs host double
    hipLaunchKernelGGL(spmv_sym14<8>, dim3(batch), dim3(512), lds, 0, M[NB - 1], x, y, N);
    CK(hipGetLastError());
    if (gbdpcg_spmv_f32(hd, n, N, batch, M[NB - 1], x, y2, nullptr) != GBDPCG_OK) { printf("spmv failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hy(velems), hy2(velems);
    CK(hipMemcpy(hy.data(), y, velems * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy2.data(), y2, velems * 4, hipMemcpyDeviceToHost));
    double num = 0, num2 = 0, den = 0;
    for (size_t p : {(size_t)0, (size_t)batch / 2, (size_t)batch - 1})
        for (int k = 0; k < N; ++k)
            for (int r = 0; r < n; ++r) {
                double acc = 0;
                const float* blk = &h[(p * N + k) * 3 * NN];
                for (int b = 0; b < 3; ++b) {
                    const int kc = k + b - 1;
                    if (kc < 0 || kc >= N) continue;
                    for (int c = 0; c < n; ++c) acc += (double)blk[b * NN + c * n + r] * hx[(p * N + kc) * n + c];
                }
                const size_t i = (p * N + k) * n + r;
                num += (hy[i] - acc) * (hy[i] - acc); num2 += (hy2[i] - acc) * (hy2[i] - acc); den += acc * acc;
            }
    printf("rel. error vs host fp64: symmetric kernel %.3e, library kernel %.3e\n", sqrt(num / den), sqrt(num2 / den));
    if (getenv("SYM_DEBUG")) {
        for (int k = 0; k < std::min(N, 3); ++k) {
            printf("row %d sym:", k); for (int r = 0; r < n; ++r) printf(" %8.4f", hy[k * n + r]); printf("\n");
            printf("row %d lib:", k); for (int r = 0; r < n; ++r) printf(" %8.4f", hy2[k * n + r]); printf("\n");
        }
    }

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t[3];
    int turn = 0;
    for (int r = 0; r < reps + 3; ++r)
        for (int v = 0; v < 3; ++v) {
            const float* Mb = M[turn++ % NB];
            CK(hipEventRecord(e0, 0));
            if (v == 0) { if (gbdpcg_spmv_f32(hd, n, N, batch, Mb, x, y2, nullptr) != GBDPCG_OK) return 1; }
            else if (v == 1) hipLaunchKernelGGL(spmv_sym14<8>, dim3(batch), dim3(512), lds, 0, Mb, x, y, N);
            else hipLaunchKernelGGL(spmv_sym14<4>, dim3(batch), dim3(256), lds, 0, Mb, x, y, N);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) t[v].push_back(ms);
        }
    const double alg = (double)batch * ((3.0 * N - 2) * NN + 2.0 * n * N) * 4;
    const char* nm[3] = {"library spmv (reads L,D,R)", "symmetric, 8 waves/problem", "symmetric, 4 waves/problem"};
    for (int v = 0; v < 3; ++v)
        printf("%-28s median %.4f ms -> %.0f GB/s algorithmic (full-matrix bytes)\n", nm[v], median(t[v]), alg / median(t[v]) / 1e6);
    {   // PCG-like: 50 passes alternating two matrices (S / Pinv), 512 problems resident at a time
        std::vector<float> tt;
        for (int r = 0; r < 8; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(pcg_like_sym14<8>, dim3(512), dim3(512), lds, 0, M[0], M[1], x, y, N, 50, batch);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) tt.push_back(ms);
        }
        printf("PCG-like symmetric loop (50 matrix passes, 8 waves x 512 resident): median %.4f ms = %.1f us per pass\n",
               median(tt), median(tt) * 1000 / 50);
    }
    gbdpcg_destroy(hd);
    return 0;
}
