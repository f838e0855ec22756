/*
 * oracle/pcg_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C) of the block-tridiagonal
 * PCG that A2R-Lab/GBD-PCG runs in its CUDA kernel pcg<T,n,N>
 * (/root/reference/include/pcg.cuh:54-218, helpers include/utils.cuh:9-85).
 * It is the CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under gbd-pcg_amd/ or
 * include/ links, loads or calls it; the product fails loudly without its HIP
 * library rather than falling back to this code.
 *
 * PINNING STATUS -- "parity unpinned" at the last-bit level, pinned by
 * known answers at the tolerance level:
 *   * The reference cannot be built here: it needs nvcc, the CUDA runtime and
 *     cooperative-groups headers, and the GLASS submodule, whose directory is
 *     empty in the checkout (.gitmodules:1-3, version unpinned).  No stand-ins
 *     for those were written; oracle/_ref does not exist.
 *   * The reference ships no tests, golden vectors or expected outputs.  Its
 *     only fixture is the INPUT system of examples/pcg_solve.cu:14-25 (same
 *     data in examples/pcg_solve_dp.cu:14-25).  tests/test_oracle.py checks
 *     this file against (a) a dense fp64 solve of that system, (b) the
 *     iteration counts SURVEY.md section 8c records for it (6 with Pinv = I,
 *     3 with the symmetric-stair Pinv, fp64), (c) dense fp64 solves / dense
 *     products of generated systems.
 *   * Summation order inside glass::dot / glass::reduce is not knowable from
 *     the tree; ORACLE_TREE vs sequential, and ORACLE_FMA on/off, bracket the
 *     plausible device orders.  HIP-vs-oracle parity is therefore stated as a
 *     norm-wise tolerance at equal iteration count (tests/test_gpu_parity.py),
 *     never bit-exact.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pcg_oracle.h"

#define T float
#define SUF f32
#define T_FMA fmaf
#define T_ABS fabsf
#include "pcg_oracle_impl.inc"
#undef T
#undef SUF
#undef T_FMA
#undef T_ABS

#define T double
#define SUF f64
#define T_FMA fma
#define T_ABS fabs
#include "pcg_oracle_impl.inc"
#undef T
#undef SUF
#undef T_FMA
#undef T_ABS

int oracle_version(void) { return 1; }
