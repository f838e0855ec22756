/* oracle/pcg_oracle.h -- TEST INFRASTRUCTURE ONLY (see pcg_oracle.c). */
#ifndef PCG_ORACLE_H
#define PCG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flags */
#define ORACLE_FMA 1  /* a*b+c as one fused multiply-add (nvcc's default contraction) */
#define ORACLE_TREE 2 /* halving-tree order inside dot / reduce instead of left-to-right */
#define ORACLE_DEFAULT (ORACLE_FMA | ORACLE_TREE)

#define ORACLE_DECL(T, SUF)                                                                  \
    int oracle_spmv_##SUF(uint32_t n, uint32_t N, const T *M, const T *x, T *y, int flags);  \
    int oracle_spmv_batch_##SUF(uint32_t n, uint32_t N, uint32_t batch, const T *M,          \
                                const T *x, T *y, int flags, int nthreads);                  \
    int oracle_pcg_##SUF(uint32_t n, uint32_t N, const T *S, const T *Pinv, const T *gamma,  \
                         T *lambda, T *r_out, T *p_out, T tol, uint32_t max_iter,            \
                         uint32_t *iters_out, uint8_t *max_iter_exit_out, T *eta_trace,      \
                         int flags);                                                         \
    int oracle_pcg_batch_##SUF(uint32_t n, uint32_t N, uint32_t batch, const T *S,           \
                               const T *Pinv, const T *gamma, T *lambda, T *r_out, T *p_out, \
                               T tol, uint32_t max_iter, uint32_t *iters_out,                \
                               uint8_t *max_iter_exit_out, int flags, int nthreads);

ORACLE_DECL(float, f32)
ORACLE_DECL(double, f64)
#undef ORACLE_DECL

int oracle_version(void);

#ifdef __cplusplus
}
#endif
#endif
