// One inner step of an SQP / MPC loop per graph replay, on the KKT blocks themselves (include/gbdpcg.h, gbdpcg_kkt_step_*):
//
//   once      : buffers, a handle, ONE executable graph of
//               { S, gamma, G^-1 from the KKT blocks ; Phi^-1 = symmetric stair from S ; PCG ; primal step z from lambda }
//   per step  : the caller rewrites the blocks in place (here: new gradients and constraint residuals, as a re-linearisation
//               does), replays the graph -- lambda of the previous step is the warm start -- and reads z when it needs it
//
// The reference tree has no code for the steps either side of the solve (/root/reference/README.md:2-11 states the system
// they produce); the convention is the one written out in include/gbdpcg.h.  This driver builds random well-posed problems
//     minimise sum_k 1/2 x_k'Q_k x_k + q_k'x_k + 1/2 u_k'R_k u_k + r_k'u_k   s.t.  x_0 = c_0,  x_{k+1} - A_k x_k - B_k u_k = c_{k+1}
// on the host and checks, for one problem per step and in double precision on the host, the two KKT residuals of what came
// back:  |G z + g + C'lambda| / |g|  (stationarity)  and  |C z - c| / |c|  (feasibility).
// usage: kkt_step_loop [batch=1024] [knotPoints=128] [steps=5]      (stateSize 14, controlSize 7, fp32)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gbdpcg.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)
#define GK(x)                                                                                        \
    do {                                                                                             \
        gbdpcg_status s_ = (x);                                                                      \
        if (s_ != GBDPCG_OK) {                                                                       \
            fprintf(stderr, "gbdpcg error %s at %s:%d\n", gbdpcg_status_string(s_), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

namespace {
constexpr uint32_t nx = 14, nu = 7;
constexpr uint32_t sg = nx * nx + nu * nu, sc = nx * nx + nx * nu, sv = nx + nu;

// M M' / m + I, column-major m x m
void spd(std::mt19937 &rng, uint32_t m, float *out)
{
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> a(m * m);
    for (auto &v : a) v = nd(rng);
    for (uint32_t c = 0; c < m; ++c)
        for (uint32_t r = 0; r < m; ++r) {
            double s = r == c ? 1.0 : 0.0;
            for (uint32_t q = 0; q < m; ++q) s += (double)a[q * m + r] * a[q * m + c] / m;
            out[c * m + r] = (float)s;
        }
}

struct Residuals {
    double stationarity, feasibility;
};

// KKT residuals of (z, lambda) for one problem, fp64 on the host, straight from the packed blocks.
Residuals kkt_residuals(uint32_t N, const float *G, const float *C, const float *g, const float *c, const float *z, const float *lam)
{
    double s2 = 0, g2 = 0, f2 = 0, c2 = 0;
    for (uint32_t k = 0; k < N; ++k) {
        const float *Q = G + (size_t)k * sg, *R = Q + nx * nx, *A = C + (size_t)k * sc, *B = A + nx * nx;
        const float *x = z + (size_t)k * sv, *u = x + nx, *q = g + (size_t)k * sv, *r = q + nx;
        const bool nxt = k + 1 < N;
        for (uint32_t i = 0; i < nx; ++i) {   // Q x + q + lambda_k - A' lambda_{k+1}
            double v = q[i] + lam[k * nx + i];
            for (uint32_t j = 0; j < nx; ++j) v += (double)Q[j * nx + i] * x[j];
            if (nxt)
                for (uint32_t j = 0; j < nx; ++j) v -= (double)A[i * nx + j] * lam[(k + 1) * nx + j];
            s2 += v * v;
            g2 += (double)q[i] * q[i];
        }
        if (nxt)
            for (uint32_t i = 0; i < nu; ++i) {   // R u + r - B' lambda_{k+1}
                double v = r[i];
                for (uint32_t j = 0; j < nu; ++j) v += (double)R[j * nu + i] * u[j];
                for (uint32_t j = 0; j < nx; ++j) v -= (double)B[i * nx + j] * lam[(k + 1) * nx + j];
                s2 += v * v;
                g2 += (double)r[i] * r[i];
            }
        for (uint32_t i = 0; i < nx; ++i) {   // x_k - A_{k-1} x_{k-1} - B_{k-1} u_{k-1} - c_k
            double v = (double)x[i] - c[k * nx + i];
            if (k > 0) {
                const float *Ap = C + (size_t)(k - 1) * sc, *Bp = Ap + nx * nx, *xp = z + (size_t)(k - 1) * sv, *up = xp + nx;
                for (uint32_t j = 0; j < nx; ++j) v -= (double)Ap[j * nx + i] * xp[j];
                for (uint32_t j = 0; j < nu; ++j) v -= (double)Bp[j * nx + i] * up[j];
            }
            f2 += v * v;
            c2 += (double)c[k * nx + i] * c[k * nx + i];
        }
    }
    return {std::sqrt(s2 / g2), std::sqrt(f2 / c2)};
}
}  // namespace

int main(int argc, char **argv)
{
    const uint32_t batch = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024, N = argc > 2 ? (uint32_t)atoi(argv[2]) : 128;
    const int steps = argc > 3 ? atoi(argv[3]) : 5;
    if (batch == 0 || N == 0 || steps < 1) {
        fprintf(stderr, "usage: kkt_step_loop [batch] [knotPoints] [steps]\n");
        return 2;
    }
    const size_t szG = (size_t)sg * N - nu * nu, szC = (size_t)sc * (N - 1), szg = (size_t)sv * N - nu, szc = (size_t)nx * N;
    const size_t szS = (size_t)3 * nx * nx * N;

    // a few distinct problems, repeated over the batch (host generation only)
    const uint32_t distinct = batch < 8 ? batch : 8;
    std::vector<float> hG(szG * batch), hC(szC * batch + 1), hg(szg * batch), hc(szc * batch);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (uint32_t b = 0; b < distinct; ++b) {
        float *G = hG.data() + b * szG, *C = hC.data() + b * szC;
        for (uint32_t k = 0; k < N; ++k) {
            spd(rng, nx, G + (size_t)k * sg);
            if (k + 1 < N) {
                spd(rng, nu, G + (size_t)k * sg + nx * nx);
                float *A = C + (size_t)k * sc, *B = A + nx * nx;
                for (uint32_t i = 0; i < nx * nx; ++i) A[i] = 0.3f * nd(rng) / std::sqrt((float)nx) + (i / nx == i % nx ? 1.f : 0.f);
                for (uint32_t i = 0; i < nx * nu; ++i) B[i] = nd(rng) / std::sqrt((float)nx);
            }
        }
    }
    for (uint32_t b = distinct; b < batch; ++b) {
        std::copy(hG.begin() + (b % distinct) * szG, hG.begin() + (b % distinct + 1) * szG, hG.begin() + b * szG);
        std::copy(hC.begin() + (b % distinct) * szC, hC.begin() + (b % distinct + 1) * szC, hC.begin() + b * szC);
    }

    float *dG, *dC, *dg, *dc, *dS, *dgam, *dGi, *dP, *dl, *dz;
    uint32_t *d_iters;
    uint8_t *d_flags;
    CK(hipMalloc((void **)&dG, szG * batch * 4));
    CK(hipMalloc((void **)&dC, (szC * batch + 1) * 4));
    CK(hipMalloc((void **)&dg, szg * batch * 4));
    CK(hipMalloc((void **)&dc, szc * batch * 4));
    CK(hipMalloc((void **)&dS, szS * batch * 4));
    CK(hipMalloc((void **)&dgam, szc * batch * 4));
    CK(hipMalloc((void **)&dGi, szG * batch * 4));
    CK(hipMalloc((void **)&dP, szS * batch * 4));
    CK(hipMalloc((void **)&dl, szc * batch * 4));
    CK(hipMalloc((void **)&dz, szg * batch * 4));
    CK(hipMalloc((void **)&d_iters, batch * 4));
    CK(hipMalloc((void **)&d_flags, batch));
    CK(hipMemcpy(dG, hG.data(), szG * batch * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, hC.data(), szC * batch * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dl, 0, szc * batch * 4));

    gbdpcg_handle_t h;
    GK(gbdpcg_create(&h, 0));
    gbdpcg_graph_t graph;
    GK(gbdpcg_graph_create_kkt_step_f32(h, nx, nu, N, batch, dG, dC, dg, dc, dS, dgam, dGi, dP, GBDPCG_PINV_STAIR, dl, nullptr, nullptr,
                                        1e-10f, 200, d_iters, d_flags, dz, &graph));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));

    std::vector<float> hz(szg), hl(szc);
    std::vector<uint32_t> hi(batch);
    std::vector<uint8_t> hf(batch);
    int bad = 0;
    for (int s = 0; s < steps; ++s) {
        // the re-linearisation of this step: gradients and residuals drift a little, so the previous lambda is a good start
        for (uint32_t b = 0; b < distinct; ++b) {
            for (size_t i = 0; i < szg; ++i) hg[b * szg + i] = (s == 0 ? 0.f : 0.9f * hg[b * szg + i]) + (s == 0 ? 1.f : 0.1f) * nd(rng);
            for (size_t i = 0; i < szc; ++i) hc[b * szc + i] = (s == 0 ? 0.f : 0.9f * hc[b * szc + i]) + (s == 0 ? 0.1f : 0.01f) * nd(rng);
        }
        for (uint32_t b = distinct; b < batch; ++b) {
            std::copy(hg.begin() + (b % distinct) * szg, hg.begin() + (b % distinct + 1) * szg, hg.begin() + b * szg);
            std::copy(hc.begin() + (b % distinct) * szc, hc.begin() + (b % distinct + 1) * szc, hc.begin() + b * szc);
        }
        CK(hipMemcpyAsync(dg, hg.data(), szg * batch * 4, hipMemcpyHostToDevice, stream));
        CK(hipMemcpyAsync(dc, hc.data(), szc * batch * 4, hipMemcpyHostToDevice, stream));
        CK(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        GK(gbdpcg_graph_launch(graph, stream));
        CK(hipStreamSynchronize(stream));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const uint32_t b = (uint32_t)s % batch;   // the problem checked this step
        CK(hipMemcpy(hz.data(), dz + b * szg, szg * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hl.data(), dl + b * szc, szc * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hi.data(), d_iters, batch * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), d_flags, batch, hipMemcpyDeviceToHost));
        double it = 0;
        uint32_t ran_out = 0;
        for (uint32_t i = 0; i < batch; ++i) {
            it += hi[i];
            ran_out += hf[i] != 0;
        }
        const Residuals r = kkt_residuals(N, hG.data() + b * szG, hC.data() + b * szC, hg.data() + b * szg, hc.data() + b * szc, hz.data(), hl.data());
        printf("step %d: %.3f ms for %u KKT systems, %.1f PCG iterations on average, %u ran out; problem %u: stationarity %.2e, feasibility %.2e\n",
               s, ms, batch, it / batch, ran_out, b, r.stationarity, r.feasibility);
        if (ran_out || !(r.stationarity < 1e-3) || !(r.feasibility < 1e-3)) ++bad;
    }
    gbdpcg_graph_destroy(graph);
    gbdpcg_destroy(h);
    for (void *p : {(void *)dG, (void *)dC, (void *)dg, (void *)dc, (void *)dS, (void *)dgam, (void *)dGi, (void *)dP, (void *)dl, (void *)dz,
                    (void *)d_iters, (void *)d_flags})
        (void)hipFree(p);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
