// internal.hpp -- launcher declarations shared by the translation units of libgbdpcg.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace gbdpcg {

// Device limits the launchers size against (filled once per handle).
struct DeviceInfo {
    int device = 0;
    int num_cus = 256;
    size_t lds_per_cu = 160 * 1024;   // MI355X: 160 KiB per CU, one workgroup may take all of it
    size_t lds_per_wg_max = 160 * 1024;
};

// What a kernel whose workgroups rendezvous in-kernel (pcg_cluster.hip, pcg_persist.hip) would leave in d_iters for a
// problem whose workgroups could not meet (with d_max_iter_exit = 2; lambda, r, p untouched).  Callers never see it: the
// workgroup of the problem that finishes last solves it alone inside the same launch (pcg_stream.hpp, stream_rescue);
// only variants/libgbdpcg_hooks.so can switch that off (GBDPCG_RESCUE_OFF) to show the mark to a test.
constexpr uint32_t kItersGaveUp = 0xffffffffu;

// Does this launch own problem `prob`?  (see PcgArgs::sel)
template <typename A> __device__ __forceinline__ bool pcg_takes(const A &a, uint32_t prob)
{
    if (!a.sel) return true;
    bool sym = true;
    for (uint32_t c = 0; c < a.sel_stride; ++c) sym &= a.sel[(size_t)prob * a.sel_stride + c] == 1;
    return sym == (a.want == 1);
}

// The same verdicts for up to 64 problems at once: bit j of the result = this launch owns problem first + j * step
// (j < count <= 64).  Every lane of the calling wave fetches the verdict bytes of one problem, so a workgroup that walks
// over many problems pays one memory round trip for all of them instead of one per problem; every wave of a workgroup
// that calls it gets the same mask.  All 64 lanes must be active.
template <typename A>
__device__ __forceinline__ unsigned long long pcg_takes_mask(const A &a, uint32_t first, uint32_t step, uint32_t count, uint32_t lane)
{
    if (!a.sel) return count >= 64 ? ~0ull : ((1ull << count) - 1ull);
    bool mine = false;
    if (lane < count) {
        const size_t prob = (size_t)first + (size_t)lane * step;
        bool sym = true;
        for (uint32_t c = 0; c < a.sel_stride; ++c) sym &= a.sel[prob * a.sel_stride + c] == 1;
        mine = sym == (a.want == 1);
    }
    return __ballot(mine);
}

template <typename T> struct SpmvArgs {
    const T *M;
    const T *x;
    T *y;
    uint32_t n, N, batch;
    bool symmetric = false;
};

template <typename T> struct PcgArgs {
    const T *S;
    const T *Pinv;  // nullptr => identity
    const T *gamma;
    T *lambda;
    T *r;  // nullable
    T *p;  // nullable
    T tol;
    uint32_t max_iter;
    uint32_t n, N, batch;
    uint32_t *iters;         // [batch]
    uint8_t *max_iter_exit;  // [batch], nullable
    bool symmetric = false;  // stream [D|R] only: caller's assertion, or per problem where sel says so
    // Per-problem kernel selection (symmetric AUTO mode): a launch handles problem b only when
    // (sel[b] == 1) == (want == 1): flag 1 -> the symmetric launch, anything else -> the general one, so
    // every problem is taken by exactly one of the two launches whatever the flag holds.  sel == nullptr:
    // every problem.
    const uint8_t *sel = nullptr;
    uint8_t want = 0;
    uint32_t sel_stride = 1;  // verdict bytes per problem (one per workgroup of the check kernel); the flag is their AND
    // Split path, blocking entry points only: a counter in host-visible memory that a problem bumps when
    // it converges, so that the host can stop enqueueing iteration launches (nullptr: not used).
    uint32_t *host_done = nullptr;
    // Cluster path (pcg_cluster.hip): the handle's hand-off slots (cluster_workspace_bytes; zero-filled once when the
    // handle is made, never cleared afterwards: a tag carries the launch number); nullptr: the path is not offered.
    void *cluster_ws = nullptr;
    // In-kernel rescue (pcg_stream.hpp): device memory for the vectors of the workgroup that solves a problem alone when
    // the workgroups of its cluster / persistent launch could not meet -- rescue_vec_elems(n, N) elements per cluster
    // (pcg_cluster.hip) or per problem (pcg_persist.hip).  rescue_off: hooks build only (the mark stays visible).
    void *rescue_vec = nullptr;
    bool rescue_off = false;
};

// Widest per-lane vector (in elements) usable for this block size and these base pointers:
// V in {1,2,4}, n % V == 0, V*sizeof(T) <= 16, every pointer V*sizeof(T)-aligned, n/V <= 64.
// Returns 0 when no mapping exists (n/V > 64 for every V).
template <typename T> int choose_vec(uint32_t n, const void *const *ptrs, int nptrs);

// Block sizes with a compile-time specialised kernel (everything else runs the runtime-n pipeline).
// BASELINE shapes (14, 36), the reference's example (2) and common MPC state sizes.
#define GBDPCG_SPECIALIZED_N(X) X(2) X(4) X(6) X(8) X(10) X(12) X(13) X(14) X(16) X(18) X(20) X(24) X(36)

// Widest per-lane vector a specialised kernel of block size NCT is built with (the V choose_vec
// returns for aligned pointers); other V values of that n fall back to the runtime-n kernel.
template <typename T, int NCT> constexpr int best_v()
{
    for (int V : {4, 2, 1})
        if ((size_t)V * sizeof(T) <= 16 && NCT % V == 0 && NCT / V <= 64) return V;
    return 1;
}

// ---- spmv.hip
template <typename T> hipError_t launch_spmv(const DeviceInfo &dev, const SpmvArgs<T> &a, hipStream_t s);

// ---- pcg_fused.hip : one workgroup per problem
template <typename T> size_t fused_lds_bytes(uint32_t n, uint32_t N, uint32_t waves);
template <typename T> bool fused_fits(const DeviceInfo &dev, uint32_t n, uint32_t N);
template <typename T> hipError_t launch_pcg_fused(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s);

// ---- pcg_resident.hip : both matrices register-resident, one 8-wave workgroup per problem.
// Returns false when the shape is not eligible (then nothing was launched).
template <typename T> bool resident_shape(uint32_t n, uint32_t N);  // shape handled by the resident kernel
template <typename T> void resident_prepare(uint32_t n, uint32_t N);  // runtime queries of that kernel, outside any capture
template <typename T>
bool launch_pcg_resident(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err);
// Symmetric matrices resident on one CU (pcg_resident_sym.hip): n = 14, fp32, N <= 128, a.symmetric set.
template <typename T> bool resident_sym_shape(uint32_t n, uint32_t N);
template <typename T>
bool launch_pcg_resident_sym(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err);

// ---- pcg_cluster.hip : general-storage matrices register-resident, a problem over a cluster of 2-4 workgroups (CUs)
// Workgroups per problem the cluster path would use; 0 = shape not handled (n = 14, fp32, 72 < N <= 288 only).
template <typename T> uint32_t cluster_members(uint32_t n, uint32_t N);
size_t cluster_workspace_bytes(const DeviceInfo &dev);
size_t cluster_rescue_bytes(const DeviceInfo &dev);   // PcgArgs::rescue_vec of a cluster launch
// Returns false when the launch is not eligible (then nothing was launched).
template <typename T>
bool launch_pcg_cluster(const DeviceInfo &dev, const PcgArgs<T> &a, hipStream_t s, hipError_t *err);

// ---- pcg_split.hip : many workgroups per problem, two launches per iteration
template <typename T> size_t split_workspace_bytes(uint32_t n, uint32_t N, uint32_t batch);
template <typename T>
hipError_t launch_pcg_split(const DeviceInfo &dev, const PcgArgs<T> &a, void *workspace, hipStream_t s,
                            const volatile uint32_t *host_done_poll = nullptr);

// ---- pcg_persist.hip : one large problem over many CUs in ONE persistent launch (matrices register-resident,
// in-kernel all-gather of {partial inner product, boundary knots} twice per iteration)
// Knots per workgroup the launch would use; 0 = the shape cannot run persistently on this device.
template <typename T>
uint32_t persist_knots_per_wg(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, bool one_reduction = false);
template <typename T> size_t persist_workspace_bytes(uint32_t n, uint32_t N, uint32_t batch);
template <typename T> size_t persist_rescue_bytes(uint32_t n, uint32_t N, uint32_t batch);   // PcgArgs::rescue_vec of a persistent launch
// workspace: persist_workspace_bytes, ZERO-FILLED once when allocated (epoch bases live there), never cleared again
template <typename T>
hipError_t launch_pcg_persist(const DeviceInfo &dev, const PcgArgs<T> &a, void *workspace, hipStream_t s,
                              bool one_reduction = false);   // one_reduction: the Chronopoulos-Gear form (opt-in path 4)
// (the kernels of the block sizes 22 - 28 / 30 - 36, compiled in units of their own: pcg_persist_b.hip / pcg_persist_c.hip)
template <typename T>
hipError_t launch_pcg_persist_b(const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction, uint32_t K);
template <typename T>
hipError_t launch_pcg_persist_c(const PcgArgs<T> &a, void *workspace, hipStream_t s, bool one_reduction, uint32_t K);

// ---- symcheck.hip : flags[b] = 1 iff L_{k+1} == R_k^T bit for bit for every k of problem b
// and_into: flags[b] &= result instead of flags[b] = result (second matrix of a pair).
template <typename T>
hipError_t launch_check_symmetric(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *M,
                                  uint8_t *flags, bool and_into, hipStream_t s);
// Device fills as kernels (capturable, arguments travel with the graph node; see symcheck.hip).
hipError_t launch_fill_bytes(uint8_t *p, uint8_t v, size_t count, hipStream_t s);
hipError_t launch_fill_words(uint32_t *p, uint32_t v, size_t count, hipStream_t s);
// S and Pinv in one launch (flags = 1 where both are symmetric); false if the shape does not fit.
template <typename T>
bool launch_check_symmetric_pair(uint32_t n, uint32_t N, uint32_t batch, const T *A, const T *B, uint8_t *flags,
                                 hipStream_t s, hipError_t *err, uint32_t *verdicts_per_problem = nullptr);
// Verdict bytes per problem the pair kernel writes when asked for per-workgroup verdicts (0: shape not supported).
template <typename T> uint32_t check_pair_chunks(uint32_t n, uint32_t N);
// Does launch_pcg_fused have a symmetric-streaming kernel for this shape (and would it be used)?
template <typename T> bool fused_has_symmetric(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch);

// ---- pinv.hip
// verdicts (optional, only when pinv_verdict_chunks() != 0): pinv_verdict_chunks bytes per problem, 1 = every pair of
// that chunk was exactly symmetric in S and therefore is in Pinv (pcg_takes ANDs them).
template <typename T>
hipError_t launch_form_pinv(const DeviceInfo &dev, uint32_t n, uint32_t N, uint32_t batch, const T *S,
                            T *Pinv, int kind, hipStream_t s, uint8_t *verdicts = nullptr, bool s_symmetric = false);
// (s_symmetric: the caller KNOWS L_{k+1} == R_k^T in S bit for bit -- the one-launch stair kernel then never reads L)
template <typename T> uint32_t pinv_verdict_chunks(uint32_t n, uint32_t N, int kind);

// ---- schur.hip: KKT blocks -> S, gamma, G^-1;  lambda -> primal step.  Layouts in include/gbdpcg.h.
template <typename T>
hipError_t launch_form_schur(const DeviceInfo &dev, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *G, const T *C,
                             const T *g, const T *c, T *S, T *gamma, T *Ginv, hipStream_t s);
template <typename T>
hipError_t launch_recover_primal(const DeviceInfo &dev, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *Ginv,
                                 const T *C, const T *g, const T *lambda, T *z, hipStream_t s);
template <typename T> bool schur_shape_ok(const DeviceInfo &dev, uint32_t nx, uint32_t nu);

}  // namespace gbdpcg
