// Synthetic symmetric positive definite block-tridiagonal systems for the example drivers, built on the host:
// S = G W G^T with a block-bidiagonal G (the structure of an MPC Schur complement), stored in the reference's
// [L_k | D_k | R_k] column-major layout (/root/reference/include/pcg.cuh:104-110) with L_{k+1} == R_k^T bit for bit.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

static constexpr uint32_t n = 14;  // stateSize of the reference's iiwa example

static double urand(uint64_t &s)  // splitmix64 -> (-1, 1)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 4503599627370496.0 - 1.0;
}

// S = G W G^T, G = I on the block diagonal and -A_k below it, W_k = I + M_k M_k^T; layout [L_k | D_k | R_k], column-major blocks
static void make_problem(uint32_t N, uint64_t seed, float *S, float *gamma)
{
    std::vector<double> A((size_t)N * n * n), W((size_t)N * n * n), tmp(n * n), D(n * n), O(n * n);
    for (uint32_t k = 0; k < N; ++k) {
        double M[n * n];
        for (double &m : M) m = urand(seed) / std::sqrt((double)n);
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                double w = r == c ? 1.0 : 0.0;
                for (uint32_t q = 0; q < n; ++q) w += M[r * n + q] * M[c * n + q];
                W[(size_t)k * n * n + r * n + c] = w;
                A[(size_t)k * n * n + r * n + c] = 0.35 * urand(seed);  // contraction: keeps S well conditioned
            }
    }
    auto at = [&](const std::vector<double> &X, uint32_t k, uint32_t r, uint32_t c) { return X[(size_t)k * n * n + r * n + c]; };
    for (uint32_t k = 0; k < N; ++k) {
        // D_k = W_k + A_k W_{k-1} A_k^T ;  O_k = S_{k,k+1} = -W_k A_{k+1}^T
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                double d = at(W, k, r, c);
                if (k > 0)
                    for (uint32_t p = 0; p < n; ++p)
                        for (uint32_t q = 0; q < n; ++q) d += at(A, k, r, p) * at(W, k - 1, p, q) * at(A, k, c, q);
                D[r * n + c] = d;
                double o = 0;
                if (k + 1 < N)
                    for (uint32_t q = 0; q < n; ++q) o -= at(W, k, r, q) * at(A, k + 1, c, q);
                O[r * n + c] = o;
            }
        float *blk = S + (size_t)k * 3 * n * n;
        for (uint32_t r = 0; r < n; ++r)
            for (uint32_t c = 0; c < n; ++c) {
                blk[n * n + c * n + r] = (float)(0.5 * (D[r * n + c] + D[c * n + r]));  // D_k, symmetrised
                blk[2 * n * n + c * n + r] = (float)O[r * n + c];                        // R_k
                if (k + 1 < N) blk[3 * n * n + r * n + c] = (float)O[r * n + c];         // L_{k+1} = R_k^T, bit for bit
            }
        if (k == 0)
            for (uint32_t i = 0; i < n * n; ++i) blk[i] = 0.f;  // L_0: never read
        if (k == N - 1)
            for (uint32_t i = 0; i < n * n; ++i) blk[2 * n * n + i] = 0.f;  // R_{N-1}: never read
        for (uint32_t r = 0; r < n; ++r) gamma[(size_t)k * n + r] = (float)urand(seed);
    }
}
