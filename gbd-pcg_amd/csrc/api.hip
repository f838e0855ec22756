// api.hip -- the C ABI of libgbdpcg.so (include/gbdpcg.h): handle, dispatch, graphs, host overloads.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/gbdpcg.h"
#include "internal.hpp"

#ifndef GBDPCG_KKT_SKIP_L
#define GBDPCG_KKT_SKIP_L 1   // 0: the stair kernel of gbdpcg_kkt_step_* reads and compares L_{k+1} like everywhere else (A/B runs)
#endif

using namespace gbdpcg;

struct gbdpcg_context {
    DeviceInfo dev;
    gbdpcg_path forced = GBDPCG_PATH_AUTO;
    int symmetric = 2;  // gbdpcg_set_symmetric: 0 never, 1 assume, 2 check on the device (default)
    uint8_t *sym_flags = nullptr;  // [sym_cap] per-problem result of the check
    size_t sym_cap = 0;
    hipError_t last_err = hipSuccess;
    // status words for the blocking entry points (replace the per-call cudaMalloc of interface.cuh:105-108)
    uint32_t *d_iters = nullptr;
    uint8_t *d_exit = nullptr;
    uint32_t *h_iters = nullptr;  // pinned
    uint8_t *h_exit = nullptr;    // pinned
    uint32_t *h_done = nullptr;   // pinned, device-visible: problems that reported convergence (blocking split solves)
    uint32_t *h_done_dev = nullptr;
    uint32_t *h_iters_dev = nullptr;  // device views of h_iters / h_exit: the blocking entry points let the kernels
    uint8_t *h_exit_dev = nullptr;    // write the two status words straight to the host (no copy-back)
    // split-path workspace, grown on demand outside capture
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // persistent path: hand-off slots and epoch bases.  One zero-filled buffer PER SHAPE (element size, n, N, batch), made
    // outside capture on first use and kept until the handle goes: the slot layout depends on the shape, and a buffer
    // that only ever sees one layout can never show a launch a stale tag left by another (graphs captured for a shape
    // keep pointing at that shape's buffer).
    struct PersistWs {
        uint64_t key;
        void *buf;
    };
    std::vector<PersistWs> pws;
    // cluster path (pcg_cluster.hip): hand-off slots, one per CU and epoch parity.  Made with the handle (its size does not
    // depend on the shape) and zero-filled once; launches never clear it (a tag carries the launch number, and the word
    // that counts launches lives in it).
    void *cluster_ws = nullptr;
    void *cluster_rescue = nullptr;   // vectors of the cluster path's in-kernel rescue (cluster_rescue_bytes)
    void *pws_last = nullptr;   // diagnostic builds only (gbdpcg_internal_persist_ws)
    // Buffers replaced by a larger one.  Graphs built earlier (gbdpcg_graph_create_solve_*, or a caller's own
    // capture of gbdpcg_solve_*) hold the OLD pointers in their kernel nodes, so growth never frees: the old
    // buffer stays valid until the handle goes.  Sizes at least double, which bounds the total at 2x the largest.
    std::vector<void *> retired;
};

struct gbdpcg_graph {
    gbdpcg_handle_t h = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

namespace gbdpcg {

template <typename T> int choose_vec(uint32_t n, const void *const *ptrs, int nptrs)
{
    const int cand[3] = {4, 2, 1};
    for (int V : cand) {
        if (V * sizeof(T) > 16 || n % V != 0 || n / V > 64) continue;
        bool ok = true;
        for (int i = 0; i < nptrs; ++i)
            if (ptrs[i] && reinterpret_cast<uintptr_t>(ptrs[i]) % (V * sizeof(T)) != 0) ok = false;
        if (ok) return V;
    }
    return 0;
}
template int choose_vec<float>(uint32_t, const void *const *, int);
template int choose_vec<double>(uint32_t, const void *const *, int);

}  // namespace gbdpcg

namespace {

gbdpcg_status fail(gbdpcg_handle_t h, hipError_t e)
{
    if (h) h->last_err = e;
    return GBDPCG_ERR_HIP;
}

#define HIP_TRY(h, expr)                                \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) return fail((h), e__);   \
    } while (0)

// Every entry point runs with the handle's device current and puts the caller's device back on return (a host
// thread that loops over the handles of several GPUs keeps whatever device it had selected).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define DEVICE_SCOPE(h)                          \
    DeviceScope device_scope__((h)->dev.device); \
    if (device_scope__.err != hipSuccess) return fail((h), device_scope__.err)

bool shape_ok(uint32_t n, uint32_t N, uint32_t batch)
{
    if (n == 0 || N == 0 || batch == 0) return false;
    // index arithmetic inside the kernels is 32-bit within one problem
    if ((uint64_t)3 * n * n * N >= (1ull << 31)) return false;
    return true;
}

template <typename T> bool mappable(uint32_t n)
{
    const void *none[1] = {nullptr};
    return choose_vec<T>(n, none, 1) != 0;
}

template <typename T> gbdpcg_path pick_path(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch)
{
    const bool fits = fused_fits<T>(h->dev, n, N);
    const bool persist = persist_knots_per_wg<T>(h->dev, n, N, batch) != 0;
    if (h->forced == GBDPCG_PATH_PERSISTENT_1R && persist_knots_per_wg<T>(h->dev, n, N, batch, true) != 0)
        return GBDPCG_PATH_PERSISTENT_1R;
    if (h->forced == GBDPCG_PATH_PERSISTENT || h->forced == GBDPCG_PATH_PERSISTENT_1R)
        return persist ? GBDPCG_PATH_PERSISTENT : (fits ? GBDPCG_PATH_FUSED : GBDPCG_PATH_SPLIT);
    if (h->forced == GBDPCG_PATH_FUSED) return fits ? GBDPCG_PATH_FUSED : GBDPCG_PATH_SPLIT;
    if (h->forced == GBDPCG_PATH_SPLIT) return GBDPCG_PATH_SPLIT;
    // A problem too large for one workgroup: one persistent launch over many CUs when all of its workgroups can be
    // resident together (measured on config 4, n=36 N=256 fp64: see DESIGN.md), else two launches per iteration.
    if (!fits) return persist ? GBDPCG_PATH_PERSISTENT : GBDPCG_PATH_SPLIT;
    // Fits one workgroup but would stream both matrices through ONE CU's share of the fabric every iteration
    // (43 GB/s measured, below): the persistent launch keeps them in registers and pays ~5.7 us per iteration.
    const bool cluster = cluster_members<T>(n, N) != 0;   // general storage resident over 2-4 CUs (pcg_cluster.hip)
    if (persist && !cluster && !resident_shape<T>(n, N) && !(h->symmetric != 0 && resident_sym_shape<T>(n, N)) &&
        6.0 * n * n * N * sizeof(T) / 43e9 > 6e-6)
        return GBDPCG_PATH_PERSISTENT;
    // Shapes whose matrices stay on the CU for the whole solve: fused, whatever the batch.  (Symmetric mode 2
    // decides per problem on the device; the path is chosen for the problems that pass.)
    if (resident_shape<T>(n, N)) return GBDPCG_PATH_FUSED;
    if (h->symmetric != 0 && resident_sym_shape<T>(n, N)) return GBDPCG_PATH_FUSED;
    if (cluster) return GBDPCG_PATH_FUSED;
    if (batch >= (uint32_t)h->dev.num_cus) return GBDPCG_PATH_FUSED;  // every CU has a problem of its own to stream
    // Streaming, fewer problems than CUs.  FUSED: one workgroup per problem pulls both matrices through ONE CU's
    // share of the fabric every iteration (43 GB/s measured on the general kernel: 13.9 us per iteration for the
    // 602 KB of n=14, N=128 fp32, at any batch below the CU count).  SPLIT: the whole chip streams, at the price of
    // two dependent launches per iteration (measured on the same shape: 7.5 us at batch 1, 7.8 at 8, 11.8 at 32,
    // 13.9 at 64 = 7.4 us + batch x bytes at ~6 TB/s).
    const double bytes = 6.0 * n * n * N * sizeof(T);
    const double t_fused = bytes / 43e9, t_split = 7.4e-6 + batch * bytes / 6e12;
    return t_split < t_fused ? GBDPCG_PATH_SPLIT : GBDPCG_PATH_FUSED;
}

// Grow *buf to at least `need` bytes (at least doubling); the old buffer is retired, not freed (see
// gbdpcg_context::retired).  Never called while a stream of this handle is capturing.
gbdpcg_status grow_buffer(gbdpcg_handle_t h, void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return GBDPCG_OK;
    size_t want = need > 2 * *cap ? need : 2 * *cap;
    want = (want + 255) / 256 * 256;
    void *fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, want);
    if (e != hipSuccess && want != (need + 255) / 256 * 256) {  // doubling was too much: exactly what is needed
        want = (need + 255) / 256 * 256;
        e = hipMalloc(&fresh, want);
    }
    if (e != hipSuccess) {
        h->last_err = e;
        return GBDPCG_ERR_ALLOC;
    }
    if (*buf) {
        try {
            h->retired.push_back(*buf);
        } catch (const std::bad_alloc &) {
            (void)hipFree(fresh);
            return GBDPCG_ERR_ALLOC;
        }
    }
    *buf = fresh;
    *cap = want;
    return GBDPCG_OK;
}

gbdpcg_status ensure_ws(gbdpcg_handle_t h, size_t bytes) { return grow_buffer(h, &h->ws, &h->ws_bytes, bytes); }

// The persistent path's workspace of one shape (see gbdpcg_context::pws): found, or -- outside capture only -- allocated
// and zero-filled (epoch bases and tags must read zero where nothing was ever written).
gbdpcg_status get_pws(gbdpcg_handle_t h, uint32_t elem, uint32_t n, uint32_t N, uint32_t batch, size_t bytes, bool may_allocate,
                      void **out)
{
    const uint64_t key = ((uint64_t)elem << 60) ^ ((uint64_t)n << 48) ^ ((uint64_t)N << 24) ^ (uint64_t)batch;
    for (const auto &e : h->pws)
        if (e.key == key) {
            *out = h->pws_last = e.buf;
            return GBDPCG_OK;
        }
    if (!may_allocate) return GBDPCG_ERR_ALLOC;
    void *buf = nullptr;
    hipError_t e = hipMalloc(&buf, bytes);
    if (e != hipSuccess) {
        h->last_err = e;
        return GBDPCG_ERR_ALLOC;
    }
    e = hipMemset(buf, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) {
        try {
            h->pws.push_back({key, buf});
        } catch (const std::bad_alloc &) {
            e = hipErrorOutOfMemory;
        }
    }
    if (e != hipSuccess) {
        (void)hipFree(buf);
        return fail(h, e);
    }
    *out = h->pws_last = buf;
    return GBDPCG_OK;
}

// Bytes of a shape's persistent-path buffer: the hand-off words, then (256-byte aligned) the vectors of the workgroup that
// solves a problem alone when its launch could not get its workgroups together (pcg_stream.hpp).
template <typename T> size_t persist_rescue_offset(uint32_t n, uint32_t N, uint32_t batch)
{
    return (persist_workspace_bytes<T>(n, N, batch) + 255) / 256 * 256;
}
template <typename T> size_t persist_total_bytes(gbdpcg_handle_t, uint32_t n, uint32_t N, uint32_t batch)
{
    return persist_rescue_offset<T>(n, N, batch) + persist_rescue_bytes<T>(n, N, batch);
}

// A few problems too many for ONE persistent launch (stateSize 20 ... 36 beyond the on-chip kernels: one or two, at short horizons
// four or five, per launch) used to fall to the split path, whose hipGraph is 2 max_iter + 4 launches whatever the iteration count
// (8 problems of 24 x 128 under max_iter = 100: 370 us of launches for nine iterations).  They are solved by a few persistent
// launches in a row instead, `persist_slices` problems each, when that row is shorter than the split path's launches alone:
// 25 us of fixed cost per persistent launch against 1.8 us per launch of the split graph.  0: no slicing.
template <typename T> uint32_t persist_slices(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, uint32_t max_iter)
{
    static const bool off = getenv("GBDPCG_NO_PERSIST_SLICES") != nullptr;   // tuning runs only
    if (off || h->forced != GBDPCG_PATH_AUTO || pick_path<T>(h, n, N, batch) != GBDPCG_PATH_SPLIT) return 0;
    uint32_t cap = 0;
    for (uint32_t b = 1; b <= 64 && b < batch && persist_knots_per_wg<T>(h->dev, n, N, b) != 0; ++b) cap = b;
    if (cap == 0) return 0;
    const uint32_t launches = (batch + cap - 1) / cap;
    if (launches > 16 || 25.0 * launches >= 1.8 * (2.0 * max_iter + 4.0)) return 0;
    return cap;
}

gbdpcg_status ensure_sym_flags(gbdpcg_handle_t h, size_t batch)
{
    return grow_buffer(h, reinterpret_cast<void **>(&h->sym_flags), &h->sym_cap, batch);
}

template <typename T>
gbdpcg_status solve_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const T *d_S, const T *d_Pinv,
                         const T *d_gamma, T *d_lambda, T *d_r, T *d_p, T tol, uint32_t max_iter,
                         uint32_t *d_iters, uint8_t *d_exit, hipStream_t stream, bool blocking = false,
                         uint32_t given_verdict_stride = 0, bool known_symmetric = false)
{
    if (!h || !d_S || !d_gamma || !d_lambda || !d_iters || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    if (!mappable<T>(n)) return GBDPCG_ERR_UNSUPPORTED;
    PcgArgs<T> a{d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, n, N, batch, d_iters, d_exit};
    a.cluster_ws = h->cluster_ws;
    a.rescue_vec = h->cluster_rescue;   // (the persistent path points it at its own buffer below)
    DEVICE_SCOPE(h);
    if (const uint32_t per = persist_slices<T>(h, n, N, batch, max_iter)) {
        const size_t ms = (size_t)3 * n * n * N, vs = (size_t)n * N;
        for (uint32_t b0 = 0; b0 < batch; b0 += per) {
            const uint32_t nb = batch - b0 < per ? batch - b0 : per;
            const gbdpcg_status st = solve_impl<T>(h, n, N, nb, d_S + b0 * ms, d_Pinv ? d_Pinv + b0 * ms : nullptr, d_gamma + b0 * vs,
                                                   d_lambda + b0 * vs, d_r ? d_r + b0 * vs : nullptr, d_p ? d_p + b0 * vs : nullptr, tol,
                                                   max_iter, d_iters + b0, d_exit ? d_exit + b0 : nullptr, stream);
            if (st != GBDPCG_OK) return st;
        }
        return GBDPCG_OK;
    }
    const gbdpcg_path path = pick_path<T>(h, n, N, batch);
    if (path == GBDPCG_PATH_PERSISTENT || path == GBDPCG_PATH_PERSISTENT_1R) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        void *pws = nullptr;
        gbdpcg_status st = get_pws(h, sizeof(T), n, N, batch, persist_total_bytes<T>(h, n, N, batch), !capturing, &pws);
        if (st != GBDPCG_OK) return st;   // capturing a shape this handle has not seen: gbdpcg_reserve first
        // The workgroups of this launch wait for each other inside the kernel; if they cannot all get onto the device
        // (somebody else's kernel holds compute units) they give up, and the one that leaves last solves the problem alone
        // with these vectors (pcg_stream.hpp).  The reference refuses such a launch before it starts (pcg.cuh:23-49).
        a.rescue_vec = static_cast<unsigned char *>(pws) + persist_rescue_offset<T>(n, N, batch);
        HIP_TRY(h, launch_pcg_persist<T>(h->dev, a, pws, stream, path == GBDPCG_PATH_PERSISTENT_1R));
    } else if (path == GBDPCG_PATH_FUSED) {
        // shapes the cluster kernel keeps resident in general storage gain nothing from symmetric STREAMING: only the
        // CU-resident symmetric kernel (N <= 128) is worth a symmetry test there
        const bool cluster_only = cluster_members<T>(n, N) != 0 && !resident_sym_shape<T>(n, N);
        bool has_sym = h->symmetric != 0 && d_Pinv != nullptr && !cluster_only && fused_has_symmetric<T>(h->dev, n, N, batch);
        // A batch that one round of clusters holds (stateSize 14, N <= 128: up to 128 problems) is not worth the test either: the
        // cluster kernel solves it in general storage in one launch while the test, the symmetric launch and the general launch
        // that finds nothing to do would still be starting (measured, 1 / 16 / 64 / 128 problems to 1e-6: 50 / 46 / 47 / 53 us
        // against 52 / 61 / 65 / 74; from 160 problems on the two are level and the symmetric kernel then pulls away).
        const uint32_t members = cluster_members<T>(n, N);
        const bool one_cluster_round = members != 0 && (uint64_t)batch * members <= (uint64_t)h->dev.num_cus &&
                                       max_iter < (1u << 18) &&   // (its hand-off tags count 2 max_iter + 4 epochs per problem in 20 bits)
                                       !(reinterpret_cast<uintptr_t>(d_S) % 8) && !(d_Pinv && reinterpret_cast<uintptr_t>(d_Pinv) % 8);
        // (in every form of mode 2 -- also where the verdicts are already known and the two paths are level --, so that
        // gbdpcg_form_pinv_solve_* and gbdpcg_kkt_step_* stay bit-identical with the separate calls; mode 1, the caller's word, keeps
        // the symmetric kernel)
        if (h->symmetric == 2 && one_cluster_round) has_sym = false;
        if (has_sym && h->symmetric == 2 && known_symmetric) {
            // the caller inside this library KNOWS that every problem is symmetric in storage (gbdpcg_kkt_step_*: S written by
            // form_schur, Pinv by the stair kernel from that S): one launch, no test, no verdict bytes, no general launch
            a.symmetric = true;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
        } else if (has_sym && h->symmetric == 2 && given_verdict_stride) {
            // the verdict bytes are already in h->sym_flags, put there on this stream by the stair kernel that just
            // formed Pinv from S (gbdpcg_form_pinv_solve_*): no test launch
            a.sel_stride = given_verdict_stride;
            a.sel = h->sym_flags;
            a.symmetric = true;
            a.want = 1;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
            a.symmetric = false;
            a.want = 0;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
        } else if (has_sym && h->symmetric == 2) {
            // AUTO: test L_{k+1} == R_k^T on the device (S, then Pinv and-ed in), then launch BOTH kernels:
            // the symmetric one takes the problems that passed, the general one the rest.  No host
            // round trip, so the whole thing stays asynchronous and graph-capturable.
            // one verdict byte per check workgroup when the pair kernel fits the shape (nothing to initialise),
            // else one flag per problem
            const uint32_t vpp = check_pair_chunks<T>(n, N);
            const bool aligned16 = !((reinterpret_cast<uintptr_t>(d_S) | reinterpret_cast<uintptr_t>(d_Pinv)) % 16);
            const size_t need_flags = (size_t)batch * (vpp && aligned16 ? vpp : 1);
            if (need_flags > h->sym_cap) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                    return GBDPCG_ERR_ALLOC;
                gbdpcg_status st = ensure_sym_flags(h, need_flags);
                if (st != GBDPCG_OK) return st;
            }
            hipError_t cerr = hipSuccess;
            uint32_t stride = 1;
            if (vpp && launch_check_symmetric_pair<T>(n, N, batch, d_S, d_Pinv, h->sym_flags, stream, &cerr, &stride)) {
                HIP_TRY(h, cerr);
            } else {
                stride = 1;
                HIP_TRY(h, launch_check_symmetric<T>(h->dev, n, N, batch, d_S, h->sym_flags, false, stream));
                HIP_TRY(h, launch_check_symmetric<T>(h->dev, n, N, batch, d_Pinv, h->sym_flags, true, stream));
            }
            a.sel_stride = stride;
            a.sel = h->sym_flags;
            a.symmetric = true;
            a.want = 1;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
            a.symmetric = false;
            a.want = 0;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
        } else {
            a.symmetric = has_sym && h->symmetric == 1;
            HIP_TRY(h, launch_pcg_fused<T>(h->dev, a, stream));
        }
    } else {
        const size_t need = split_workspace_bytes<T>(n, N, batch);
        if (need > h->ws_bytes) {
            // growing allocates and synchronises: not allowed while the stream is capturing
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return GBDPCG_ERR_ALLOC;
            gbdpcg_status st = ensure_ws(h, need);
            if (st != GBDPCG_OK) return st;
        }
        const volatile uint32_t *poll = nullptr;
        if (blocking && h->h_done_dev) {  // the caller synchronises before returning: nobody else bumps the counter
            h->h_done[0] = 0;  // problems that converged
            h->h_done[1] = 0;  // iterations the device has started
            a.host_done = h->h_done_dev;
            poll = h->h_done;
        }
        HIP_TRY(h, launch_pcg_split<T>(h->dev, a, h->ws, stream, poll));
    }
    return GBDPCG_OK;
}

template <typename T>
gbdpcg_status spmv_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const T *d_M, const T *d_x,
                        T *d_y, hipStream_t stream)
{
    if (!h || !d_M || !d_x || !d_y || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    if (!mappable<T>(n)) return GBDPCG_ERR_UNSUPPORTED;
    DEVICE_SCOPE(h);
    SpmvArgs<T> a{d_M, d_x, d_y, n, N, batch};
    a.symmetric = h->symmetric == 1;  // a check would cost as much as the product itself
    HIP_TRY(h, launch_spmv<T>(h->dev, a, stream));
    return GBDPCG_OK;
}

// Verdict bytes a (n, N, batch) solve in symmetric mode 2 may need: the larger of what the test kernel and the
// one-launch stair kernel write per problem.
template <typename T> size_t verdict_bytes(uint32_t n, uint32_t N, uint32_t batch)
{
    const uint32_t a = check_pair_chunks<T>(n, N), b = pinv_verdict_chunks<T>(n, N, GBDPCG_PINV_STAIR);
    const uint32_t m = a > b ? a : b;
    return (size_t)batch * (m ? m : 1);
}

// Phi^-1 from S, then the solve, on one stream.  When the stair kernel can report, per problem, that S was exactly
// symmetric (then so is the Pinv it wrote), the solve takes those verdicts instead of launching its own test.
template <typename T>
gbdpcg_status form_pinv_solve_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const T *d_S, T *d_Pinv,
                                   gbdpcg_pinv_kind kind, const T *d_gamma, T *d_lambda, T *d_r, T *d_p, T tol,
                                   uint32_t max_iter, uint32_t *d_iters, uint8_t *d_exit, hipStream_t stream)
{
    if (!h || !d_S || !d_Pinv || !d_gamma || !d_lambda || !d_iters || !shape_ok(n, N, batch) || (int)kind < 0 ||
        (int)kind > 2)
        return GBDPCG_ERR_INVALID;
    if (!mappable<T>(n)) return GBDPCG_ERR_UNSUPPORTED;
    DEVICE_SCOPE(h);
    uint32_t stride = 0;
    if (h->symmetric == 2 && pick_path<T>(h, n, N, batch) == GBDPCG_PATH_FUSED &&
        fused_has_symmetric<T>(h->dev, n, N, batch))
        stride = pinv_verdict_chunks<T>(n, N, (int)kind);
    if (stride) {
        const size_t need = verdict_bytes<T>(n, N, batch);
        if (need > h->sym_cap) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return GBDPCG_ERR_ALLOC;
            gbdpcg_status st = ensure_sym_flags(h, need);
            if (st != GBDPCG_OK) return st;
        }
    }
    HIP_TRY(h, launch_form_pinv<T>(h->dev, n, N, batch, d_S, d_Pinv, (int)kind, stream, stride ? h->sym_flags : nullptr));
    return solve_impl<T>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit, stream,
                         false, stride);
}

template <typename T>
gbdpcg_status solve_blocking_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, const T *d_S, const T *d_Pinv,
                                  const T *d_gamma, T *d_lambda, T *d_r, T *d_p, T tol, uint32_t max_iter,
                                  uint32_t *h_iters, uint8_t *h_exit)
{
    if (!h) return GBDPCG_ERR_INVALID;
    hipStream_t s = nullptr;  // the reference launches on the default stream (interface.cuh:132)
    // status words: written by the kernels straight into coherent pinned host memory when it is mapped
    // (saves the two copy-backs of interface.cuh:136,141), else into device words that are copied back
    const bool direct = h->h_iters_dev != nullptr;
    gbdpcg_status st = solve_impl<T>(h, n, N, 1, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter,
                                     direct ? h->h_iters_dev : h->d_iters, direct ? h->h_exit_dev : h->d_exit, s, true);
    if (st != GBDPCG_OK) return st;
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->h_iters, h->d_iters, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->h_exit, h->d_exit, sizeof(uint8_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(h, hipStreamSynchronize(s));  // interface.cuh:136 synchronises through its blocking copy
    if (h_iters) *h_iters = *h->h_iters;
    if (h_exit) *h_exit = *h->h_exit;
    return GBDPCG_OK;
}

template <typename T>
gbdpcg_status solve_host_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, const T *h_S, const T *h_Pinv,
                              const T *h_gamma, T *h_lambda, T tol, uint32_t max_iter, uint32_t *h_iters,
                              uint8_t *h_exit)
{
    if (!h || !h_S || !h_gamma || !h_lambda || !shape_ok(n, N, 1)) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    const size_t mbytes = (size_t)3 * n * n * N * sizeof(T), vbytes = (size_t)n * N * sizeof(T);
    // one allocation: S | Pinv | gamma | lambda (256-byte aligned pieces)
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t total = 2 * up(mbytes) + 2 * up(vbytes);
    unsigned char *base = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), total);
    if (e != hipSuccess) {
        h->last_err = e;
        return GBDPCG_ERR_ALLOC;
    }
    T *d_S = reinterpret_cast<T *>(base);
    T *d_P = reinterpret_cast<T *>(base + up(mbytes));
    T *d_g = reinterpret_cast<T *>(base + 2 * up(mbytes));
    T *d_l = reinterpret_cast<T *>(base + 2 * up(mbytes) + up(vbytes));
    gbdpcg_status st = GBDPCG_OK;
    auto chk = [&](hipError_t err) {
        if (err != hipSuccess && st == GBDPCG_OK) st = fail(h, err);
    };
    chk(hipMemcpy(d_S, h_S, mbytes, hipMemcpyHostToDevice));
    if (h_Pinv) chk(hipMemcpy(d_P, h_Pinv, mbytes, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_g, h_gamma, vbytes, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_l, h_lambda, vbytes, hipMemcpyHostToDevice));
    if (st == GBDPCG_OK)
        st = solve_blocking_impl<T>(h, n, N, d_S, h_Pinv ? d_P : nullptr, d_g, d_l, nullptr, nullptr, tol, max_iter,
                                    h_iters, h_exit);
    if (st == GBDPCG_OK) chk(hipMemcpy(h_lambda, d_l, vbytes, hipMemcpyDeviceToHost));
    (void)hipFree(base);
    return st;
}

template <typename T>
gbdpcg_status form_schur_impl(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *d_G, const T *d_C,
                              const T *d_g, const T *d_c, T *d_S, T *d_gamma, T *d_Ginv, void *stream);
template <typename T>
gbdpcg_status recover_primal_impl(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *d_Ginv,
                                  const T *d_C, const T *d_g, const T *d_lambda, T *d_z, void *stream);

// The operands of the steps either side of the solve (gbdpcg_kkt_step_*): S and gamma are formed from them before, z after.
template <typename T> struct KktOperands {
    uint32_t nu;
    const T *G, *C, *g, *c;
    T *Ginv, *z;
};

// KKT blocks -> S, gamma, G^-1 -> Phi^-1 -> PCG -> primal step, on one stream (capturable: no allocation after the first
// call of a shape, no synchronisation).
template <typename T>
gbdpcg_status kkt_step_impl(gbdpcg_handle_t h, uint32_t nx, uint32_t N, uint32_t batch, const KktOperands<T> &k, T *d_S, T *d_gamma,
                            T *d_Pinv, gbdpcg_pinv_kind kind, T *d_lambda, T *d_r, T *d_p, T tol, uint32_t max_iter,
                            uint32_t *d_iters, uint8_t *d_exit, hipStream_t stream)
{
    if (!k.Ginv || !k.z) return GBDPCG_ERR_INVALID;
    gbdpcg_status st = form_schur_impl<T>(h, nx, k.nu, N, batch, k.G, k.C, k.g, k.c, d_S, d_gamma, k.Ginv, stream);
    if (st != GBDPCG_OK) return st;
    // form_schur writes S exactly symmetric in storage (R_k and L_{k+1} are copies of the same registers), and where the
    // one-launch stair kernel forms Pinv from such an S it writes every pair as mirror images: in the default symmetric mode
    // the verdicts of gbdpcg_form_pinv_solve_* would all read "symmetric", so they are neither written nor read here and the
    // general-storage launch that would own nothing (11 us of a 0.66 ms step) is not made.  Same kernels on the same numbers:
    // same results as the three calls.
    bool known = false;
    {
        DEVICE_SCOPE(h);
        known = d_S && d_Pinv && shape_ok(nx, N, batch) && mappable<T>(nx) && (int)kind >= 0 && (int)kind <= 2 && h->symmetric == 2 &&
                pick_path<T>(h, nx, N, batch) == GBDPCG_PATH_FUSED && fused_has_symmetric<T>(h->dev, nx, N, batch) &&
                pinv_verdict_chunks<T>(nx, N, (int)kind) != 0;
    }
    if (known) {
        if (!d_gamma || !d_lambda || !d_iters) return GBDPCG_ERR_INVALID;
        DEVICE_SCOPE(h);
        HIP_TRY(h, launch_form_pinv<T>(h->dev, nx, N, batch, d_S, d_Pinv, (int)kind, stream, nullptr, GBDPCG_KKT_SKIP_L != 0));
        st = solve_impl<T>(h, nx, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit, stream, false,
                           0, true);
    } else {
        st = form_pinv_solve_impl<T>(h, nx, N, batch, d_S, d_Pinv, kind, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit,
                                     stream);
    }
    if (st != GBDPCG_OK) return st;
    return recover_primal_impl<T>(h, nx, k.nu, N, batch, k.Ginv, k.C, k.g, d_lambda, k.z, stream);
}

template <typename T>
gbdpcg_status graph_create_impl(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const T *d_S,
                                const T *d_Pinv, const T *d_gamma, T *d_lambda, T *d_r, T *d_p, T tol,
                                uint32_t max_iter, uint32_t *d_iters, uint8_t *d_exit, gbdpcg_graph_t *out,
                                int form_kind = -1,  // >= 0: the graph also forms Pinv (written through d_Pinv) from S
                                const KktOperands<T> *kkt = nullptr)  // the graph also forms S, gamma (written through d_S, d_gamma) and recovers z
{
    if (!h || !out) return GBDPCG_ERR_INVALID;
    *out = nullptr;
    if (!shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    if (const uint32_t per = persist_slices<T>(h, n, N, batch, max_iter)) {   // the hand-off words of the slices (solve_impl)
        void *pws = nullptr;
        gbdpcg_status st = get_pws(h, sizeof(T), n, N, per, persist_total_bytes<T>(h, n, N, per), true, &pws);
        if (st == GBDPCG_OK && batch % per) st = get_pws(h, sizeof(T), n, N, batch % per, persist_total_bytes<T>(h, n, N, batch % per), true, &pws);
        if (st != GBDPCG_OK) return st;
    } else if (pick_path<T>(h, n, N, batch) == GBDPCG_PATH_PERSISTENT || pick_path<T>(h, n, N, batch) == GBDPCG_PATH_PERSISTENT_1R) {
        void *pws = nullptr;
        gbdpcg_status st = get_pws(h, sizeof(T), n, N, batch, persist_total_bytes<T>(h, n, N, batch), true, &pws);
        if (st != GBDPCG_OK) return st;
    } else if (pick_path<T>(h, n, N, batch) == GBDPCG_PATH_SPLIT) {
        gbdpcg_status st = ensure_ws(h, split_workspace_bytes<T>(n, N, batch));
        if (st != GBDPCG_OK) return st;
    } else if (h->symmetric == 2) {
        gbdpcg_status st = ensure_sym_flags(h, verdict_bytes<T>(n, N, batch));
        if (st != GBDPCG_OK) return st;
    }
    resident_prepare<T>(n, N);
    hipStream_t cs = nullptr;
    HIP_TRY(h, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    gbdpcg_graph *g = new (std::nothrow) gbdpcg_graph;
    if (!g) {
        (void)hipStreamDestroy(cs);
        return GBDPCG_ERR_ALLOC;
    }
    g->h = h;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    gbdpcg_status st = GBDPCG_OK;
    if (e == hipSuccess) {
        if (kkt)
            st = kkt_step_impl<T>(h, n, N, batch, *kkt, const_cast<T *>(d_S), const_cast<T *>(d_gamma), const_cast<T *>(d_Pinv),
                                  (gbdpcg_pinv_kind)form_kind, d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit, cs);
        else if (form_kind >= 0)
            st = form_pinv_solve_impl<T>(h, n, N, batch, d_S, const_cast<T *>(d_Pinv), (gbdpcg_pinv_kind)form_kind, d_gamma,
                                         d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit, cs);
        else
            st = solve_impl<T>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters, d_exit,
                               cs);
        hipError_t e2 = hipStreamEndCapture(cs, &g->graph);
        if (st == GBDPCG_OK && e2 != hipSuccess) st = fail(h, e2);
    } else {
        st = fail(h, e);
    }
    if (st == GBDPCG_OK) {
        e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) st = fail(h, e);
    }
    (void)hipStreamDestroy(cs);
    if (st != GBDPCG_OK) {
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return st;
    }
    *out = g;
    return GBDPCG_OK;
}

template <typename T>
gbdpcg_status csr_to_bt_impl(uint32_t n, uint32_t N, const uint32_t *row_ptr, const uint32_t *col_ind,
                             const T *val, T *h_M)
{
    if (!row_ptr || !col_ind || !val || !h_M || n == 0 || N == 0) return GBDPCG_ERR_INVALID;
    const size_t total = (size_t)3 * n * n * N;
    for (size_t i = 0; i < total; ++i) h_M[i] = T(0);
    const uint32_t rows = n * N;
    for (uint32_t row = 0; row < rows; ++row) {
        const uint32_t k = row / n, r = row - k * n;
        for (uint32_t q = row_ptr[row]; q < row_ptr[row + 1]; ++q) {
            const uint32_t col = col_ind[q];
            if (col >= rows) return GBDPCG_ERR_INVALID;
            const uint32_t kc = col / n, c = col - kc * n;
            if (kc + 1 < k || kc > k + 1) {
                if (val[q] != T(0)) return GBDPCG_ERR_INVALID;  // outside the block-tridiagonal pattern
                continue;
            }
            const uint32_t b = kc + 1 - k;  // 0: L, 1: D, 2: R
            h_M[(size_t)k * 3 * n * n + (size_t)b * n * n + (size_t)c * n + r] += val[q];
        }
    }
    return GBDPCG_OK;
}

}  // namespace

namespace {
template <typename T>
gbdpcg_status form_schur_impl(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *d_G, const T *d_C,
                              const T *d_g, const T *d_c, T *d_S, T *d_gamma, T *d_Ginv, void *stream)
{
    if (!h || !d_G || !d_g || !d_c || !d_S || !d_gamma || (!d_C && N > 1) || nu == 0 || !shape_ok(nx, N, batch))
        return GBDPCG_ERR_INVALID;
    if (!schur_shape_ok<T>(h->dev, nx, nu)) return GBDPCG_ERR_UNSUPPORTED;
    DEVICE_SCOPE(h);
    HIP_TRY(h, launch_form_schur<T>(h->dev, nx, nu, N, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, (hipStream_t)stream));
    return GBDPCG_OK;
}
template <typename T>
gbdpcg_status recover_primal_impl(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const T *d_Ginv,
                                  const T *d_C, const T *d_g, const T *d_lambda, T *d_z, void *stream)
{
    if (!h || !d_Ginv || !d_g || !d_lambda || !d_z || (!d_C && N > 1) || nu == 0 || !shape_ok(nx, N, batch))
        return GBDPCG_ERR_INVALID;
    if (!schur_shape_ok<T>(h->dev, nx, nu)) return GBDPCG_ERR_UNSUPPORTED;
    DEVICE_SCOPE(h);
    HIP_TRY(h, launch_recover_primal<T>(h->dev, nx, nu, N, batch, d_Ginv, d_C, d_g, d_lambda, d_z, (hipStream_t)stream));
    return GBDPCG_OK;
}
}  // namespace

extern "C" {

gbdpcg_status gbdpcg_create(gbdpcg_handle_t *out, int device)
{
    if (!out) return GBDPCG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return GBDPCG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GBDPCG_ERR_NO_DEVICE;
    // this library carries gfx950 code objects only
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GBDPCG_ERR_NO_DEVICE;
    gbdpcg_context *h = new (std::nothrow) gbdpcg_context;
    if (!h) return GBDPCG_ERR_ALLOC;
    h->dev.device = device;
    h->dev.num_cus = prop.multiProcessorCount;
    // gfx950: 160 KiB of LDS per CU, and one workgroup may take all of it (hipDeviceProp_t reports the
    // 64 KiB default window; larger dynamic sizes are opted into per kernel with hipFuncSetAttribute)
    h->dev.lds_per_cu = 160 * 1024;
    h->dev.lds_per_wg_max = 160 * 1024;
    DeviceScope scope(device);
    hipError_t e = scope.err;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h->d_iters), 256);
    if (e == hipSuccess) e = hipMalloc(&h->cluster_ws, cluster_workspace_bytes(h->dev));
    if (e == hipSuccess) e = hipMemset(h->cluster_ws, 0, cluster_workspace_bytes(h->dev));
    if (e == hipSuccess) e = hipMalloc(&h->cluster_rescue, cluster_rescue_bytes(h->dev));
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the fill is done before any stream of the caller can use the handle
    if (e == hipSuccess) {
        h->d_exit = reinterpret_cast<uint8_t *>(h->d_iters) + 128;
        // coherent (fine-grained) mapping: the device-side bump of h_done must reach the host while the stream still runs
        e = hipHostMalloc(reinterpret_cast<void **>(&h->h_iters), 256, hipHostMallocCoherent | hipHostMallocMapped);
    }
    if (e != hipSuccess) {
        if (h->d_iters) (void)hipFree(h->d_iters);
        if (h->cluster_ws) (void)hipFree(h->cluster_ws);
        if (h->cluster_rescue) (void)hipFree(h->cluster_rescue);
        delete h;
        return GBDPCG_ERR_HIP;
    }
    h->h_exit = reinterpret_cast<uint8_t *>(h->h_iters) + 128;
    h->h_done = h->h_iters + 16;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h->h_done_dev), h->h_done, 0) != hipSuccess) h->h_done_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&h->h_iters_dev), h->h_iters, 0) == hipSuccess)
        h->h_exit_dev = reinterpret_cast<uint8_t *>(h->h_iters_dev) + 128;
    else
        h->h_iters_dev = nullptr;
    *out = h;
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_destroy(gbdpcg_handle_t h)
{
    if (!h) return GBDPCG_ERR_INVALID;
    DeviceScope scope(h->dev.device);
    if (h->ws) (void)hipFree(h->ws);
    for (const auto &e : h->pws) (void)hipFree(e.buf);
    if (h->sym_flags) (void)hipFree(h->sym_flags);
    if (h->cluster_ws) (void)hipFree(h->cluster_ws);
    if (h->cluster_rescue) (void)hipFree(h->cluster_rescue);
    for (void *old : h->retired) (void)hipFree(old);
    if (h->d_iters) (void)hipFree(h->d_iters);
    if (h->h_iters) (void)hipHostFree(h->h_iters);
    delete h;
    return GBDPCG_OK;
}

const char *gbdpcg_status_string(gbdpcg_status s)
{
    switch (s) {
    case GBDPCG_OK: return "ok";
    case GBDPCG_ERR_INVALID: return "invalid argument";
    case GBDPCG_ERR_HIP: return "HIP runtime error";
    case GBDPCG_ERR_NO_DEVICE: return "no usable gfx950 device";
    case GBDPCG_ERR_UNSUPPORTED: return "unsupported shape";
    case GBDPCG_ERR_TOO_LARGE: return "problem does not fit on the device";
    case GBDPCG_ERR_ALLOC: return "allocation failed (or workspace growth requested during stream capture)";
    case GBDPCG_ERR_NOT_IMPLEMENTED: return "not implemented";
    }
    return "unknown status";
}

int gbdpcg_last_hip_error(gbdpcg_handle_t h) { return h ? (int)h->last_err : (int)hipErrorInvalidValue; }
const char *gbdpcg_last_hip_error_string(gbdpcg_handle_t h)
{
    return hipGetErrorString(h ? h->last_err : hipErrorInvalidValue);
}

gbdpcg_status gbdpcg_set_path(gbdpcg_handle_t h, gbdpcg_path path)
{
    if (!h || (int)path < 0 || (int)path > 4) return GBDPCG_ERR_INVALID;
    h->forced = path;
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_set_symmetric(gbdpcg_handle_t h, int mode)
{
    if (!h || mode < 0 || mode > 2) return GBDPCG_ERR_INVALID;
    h->symmetric = mode;
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_check_symmetric_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const float *d_M,
                                         uint8_t *d_flags, void *stream)
{
    if (!h || !d_M || !d_flags || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    hipError_t cerr = hipSuccess;
    if (launch_check_symmetric_pair<float>(n, N, batch, d_M, nullptr, d_flags, (hipStream_t)stream, &cerr)) {
        HIP_TRY(h, cerr);
        return GBDPCG_OK;
    }
    HIP_TRY(h, launch_check_symmetric<float>(h->dev, n, N, batch, d_M, d_flags, false, (hipStream_t)stream));
    return GBDPCG_OK;
}
gbdpcg_status gbdpcg_check_symmetric_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const double *d_M,
                                         uint8_t *d_flags, void *stream)
{
    if (!h || !d_M || !d_flags || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    hipError_t cerr = hipSuccess;
    if (launch_check_symmetric_pair<double>(n, N, batch, d_M, nullptr, d_flags, (hipStream_t)stream, &cerr)) {
        HIP_TRY(h, cerr);
        return GBDPCG_OK;
    }
    HIP_TRY(h, launch_check_symmetric<double>(h->dev, n, N, batch, d_M, d_flags, false, (hipStream_t)stream));
    return GBDPCG_OK;
}

gbdpcg_path gbdpcg_choose_path(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N, uint32_t batch)
{
    if (!h || !shape_ok(n, N, batch)) return GBDPCG_PATH_AUTO;
    return elem_size == 8 ? pick_path<double>(h, n, N, batch) : pick_path<float>(h, n, N, batch);
}

uint32_t gbdpcg_cluster_members(uint32_t elem_size, uint32_t n, uint32_t N)
{
    if (!shape_ok(n, N, 1)) return 0;
    return elem_size == 8 ? cluster_members<double>(n, N) : cluster_members<float>(n, N);
}

size_t gbdpcg_pcg_shared_mem_size(uint32_t elem_size, uint32_t n, uint32_t N)
{
    const size_t nn = (size_t)n * n, mx = n > N ? n : N;
    const size_t a = 6 * nn + 10 * (size_t)n + 2 * mx, b = 9 * nn;
    return elem_size * (a > b ? a : b);
}

gbdpcg_status gbdpcg_check_occupancy(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N, uint32_t batch)
{
    if (!h || (elem_size != 4 && elem_size != 8) || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    if (!(elem_size == 8 ? mappable<double>(n) : mappable<float>(n))) return GBDPCG_ERR_UNSUPPORTED;
    // FUSED needs one workgroup's LDS; SPLIT needs its workspace in HBM and one (rpw=1) window in LDS.
    const bool fits = elem_size == 8 ? fused_fits<double>(h->dev, n, N) : fused_fits<float>(h->dev, n, N);
    if (fits) return GBDPCG_OK;
    if ((size_t)3 * n * elem_size + 64 * elem_size > h->dev.lds_per_wg_max) return GBDPCG_ERR_TOO_LARGE;
    size_t free_b = 0, total_b = 0;
    DEVICE_SCOPE(h);
    HIP_TRY(h, hipMemGetInfo(&free_b, &total_b));
    const size_t need = gbdpcg_workspace_bytes(h, elem_size, n, N, batch);
    return need <= h->ws_bytes || need <= free_b ? GBDPCG_OK : GBDPCG_ERR_TOO_LARGE;  // growth keeps the old buffer
}

size_t gbdpcg_workspace_bytes(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N, uint32_t batch)
{
    if (!h || !shape_ok(n, N, batch)) return 0;
    if (elem_size == 8) {
        const gbdpcg_path p = pick_path<double>(h, n, N, batch);
        if (p == GBDPCG_PATH_FUSED) return 0;
        return (p == GBDPCG_PATH_PERSISTENT || p == GBDPCG_PATH_PERSISTENT_1R) ? persist_total_bytes<double>(h, n, N, batch) : split_workspace_bytes<double>(n, N, batch);
    }
    const gbdpcg_path p = pick_path<float>(h, n, N, batch);
    if (p == GBDPCG_PATH_FUSED) return 0;
    return (p == GBDPCG_PATH_PERSISTENT || p == GBDPCG_PATH_PERSISTENT_1R) ? persist_total_bytes<float>(h, n, N, batch) : split_workspace_bytes<float>(n, N, batch);
}

gbdpcg_status gbdpcg_reserve(gbdpcg_handle_t h, uint32_t elem_size, uint32_t n, uint32_t N, uint32_t batch)
{
    if (!h || (elem_size != 4 && elem_size != 8) || !shape_ok(n, N, batch)) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    // the verdict bytes of the device symmetry check (mode 2) must exist before a capture as well
    gbdpcg_status st = ensure_sym_flags(h, elem_size == 8 ? verdict_bytes<double>(n, N, batch) : verdict_bytes<float>(n, N, batch));
    if (st != GBDPCG_OK) return st;
    const gbdpcg_path p = elem_size == 8 ? pick_path<double>(h, n, N, batch) : pick_path<float>(h, n, N, batch);
    if (p == GBDPCG_PATH_PERSISTENT || p == GBDPCG_PATH_PERSISTENT_1R) {
        void *pws = nullptr;
        return get_pws(h, elem_size, n, N, batch, gbdpcg_workspace_bytes(h, elem_size, n, N, batch), true, &pws);
    }
    // a batch that a later solve may cut into persistent launches (persist_slices: whether it does depends on that call's max_iter):
    // the slices' hand-off words as well as the split path's workspace
    const uint32_t per = elem_size == 8 ? persist_slices<double>(h, n, N, batch, 0x7fffffffu) : persist_slices<float>(h, n, N, batch, 0x7fffffffu);
    if (per) {
        void *pws = nullptr;
        for (uint32_t nb : {per, batch % per}) {
            if (nb == 0) continue;
            st = get_pws(h, elem_size, n, N, nb, elem_size == 8 ? persist_total_bytes<double>(h, n, N, nb) : persist_total_bytes<float>(h, n, N, nb), true, &pws);
            if (st != GBDPCG_OK) return st;
        }
    }
    return ensure_ws(h, gbdpcg_workspace_bytes(h, elem_size, n, N, batch));
}

gbdpcg_status gbdpcg_spmv_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const float *d_M,
                              const float *d_x, float *d_y, void *stream)
{
    return spmv_impl<float>(h, n, N, batch, d_M, d_x, d_y, (hipStream_t)stream);
}
gbdpcg_status gbdpcg_spmv_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const double *d_M,
                              const double *d_x, double *d_y, void *stream)
{
    return spmv_impl<double>(h, n, N, batch, d_M, d_x, d_y, (hipStream_t)stream);
}

gbdpcg_status gbdpcg_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const float *d_S,
                               const float *d_Pinv, const float *d_gamma, float *d_lambda, float *d_r, float *d_p,
                               float tol, uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit,
                               void *stream)
{
    return solve_impl<float>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters,
                             d_max_iter_exit, (hipStream_t)stream);
}
gbdpcg_status gbdpcg_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const double *d_S,
                               const double *d_Pinv, const double *d_gamma, double *d_lambda, double *d_r,
                               double *d_p, double tol, uint32_t max_iter, uint32_t *d_iters,
                               uint8_t *d_max_iter_exit, void *stream)
{
    return solve_impl<double>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters,
                              d_max_iter_exit, (hipStream_t)stream);
}

gbdpcg_status gbdpcg_solve_blocking_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, const float *d_S,
                                        const float *d_Pinv, const float *d_gamma, float *d_lambda, float *d_r,
                                        float *d_p, float tol, uint32_t max_iter, uint32_t *h_iters,
                                        uint8_t *h_max_iter_exit)
{
    return solve_blocking_impl<float>(h, n, N, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, h_iters,
                                      h_max_iter_exit);
}
gbdpcg_status gbdpcg_solve_blocking_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, const double *d_S,
                                        const double *d_Pinv, const double *d_gamma, double *d_lambda, double *d_r,
                                        double *d_p, double tol, uint32_t max_iter, uint32_t *h_iters,
                                        uint8_t *h_max_iter_exit)
{
    return solve_blocking_impl<double>(h, n, N, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, h_iters,
                                       h_max_iter_exit);
}

gbdpcg_status gbdpcg_solve_host_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, const float *h_S,
                                    const float *h_Pinv, const float *h_gamma, float *h_lambda, float tol,
                                    uint32_t max_iter, uint32_t *h_iters, uint8_t *h_max_iter_exit)
{
    return solve_host_impl<float>(h, n, N, h_S, h_Pinv, h_gamma, h_lambda, tol, max_iter, h_iters, h_max_iter_exit);
}
gbdpcg_status gbdpcg_solve_host_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, const double *h_S,
                                    const double *h_Pinv, const double *h_gamma, double *h_lambda, double tol,
                                    uint32_t max_iter, uint32_t *h_iters, uint8_t *h_max_iter_exit)
{
    return solve_host_impl<double>(h, n, N, h_S, h_Pinv, h_gamma, h_lambda, tol, max_iter, h_iters,
                                   h_max_iter_exit);
}

gbdpcg_status gbdpcg_graph_create_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                            const float *d_S, const float *d_Pinv, const float *d_gamma,
                                            float *d_lambda, float *d_r, float *d_p, float tol, uint32_t max_iter,
                                            uint32_t *d_iters, uint8_t *d_max_iter_exit, gbdpcg_graph_t *out)
{
    return graph_create_impl<float>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter,
                                    d_iters, d_max_iter_exit, out);
}
gbdpcg_status gbdpcg_graph_create_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                            const double *d_S, const double *d_Pinv, const double *d_gamma,
                                            double *d_lambda, double *d_r, double *d_p, double tol,
                                            uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit,
                                            gbdpcg_graph_t *out)
{
    return graph_create_impl<double>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter,
                                     d_iters, d_max_iter_exit, out);
}

gbdpcg_status gbdpcg_graph_launch(gbdpcg_graph_t g, void *stream)
{
    if (!g || !g->exec) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(g->h);
    HIP_TRY(g->h, hipGraphLaunch(g->exec, (hipStream_t)stream));
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_graph_destroy(gbdpcg_graph_t g)
{
    if (!g) return GBDPCG_ERR_INVALID;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_form_pinv_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const float *d_S,
                                   float *d_Pinv, gbdpcg_pinv_kind kind, void *stream)
{
    if (!h || !d_S || !d_Pinv || !shape_ok(n, N, batch) || (int)kind < 0 || (int)kind > 2) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    HIP_TRY(h, launch_form_pinv<float>(h->dev, n, N, batch, d_S, d_Pinv, (int)kind, (hipStream_t)stream));
    return GBDPCG_OK;
}
gbdpcg_status gbdpcg_form_pinv_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const double *d_S,
                                   double *d_Pinv, gbdpcg_pinv_kind kind, void *stream)
{
    if (!h || !d_S || !d_Pinv || !shape_ok(n, N, batch) || (int)kind < 0 || (int)kind > 2) return GBDPCG_ERR_INVALID;
    DEVICE_SCOPE(h);
    HIP_TRY(h, launch_form_pinv<double>(h->dev, n, N, batch, d_S, d_Pinv, (int)kind, (hipStream_t)stream));
    return GBDPCG_OK;
}

gbdpcg_status gbdpcg_form_schur_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const float *d_G,
                                    const float *d_C, const float *d_g, const float *d_c, float *d_S, float *d_gamma,
                                    float *d_Ginv, void *stream)
{
    return form_schur_impl<float>(h, nx, nu, N, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, stream);
}
gbdpcg_status gbdpcg_form_schur_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const double *d_G,
                                    const double *d_C, const double *d_g, const double *d_c, double *d_S, double *d_gamma,
                                    double *d_Ginv, void *stream)
{
    return form_schur_impl<double>(h, nx, nu, N, batch, d_G, d_C, d_g, d_c, d_S, d_gamma, d_Ginv, stream);
}
gbdpcg_status gbdpcg_recover_primal_f32(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                        const float *d_Ginv, const float *d_C, const float *d_g, const float *d_lambda,
                                        float *d_z, void *stream)
{
    return recover_primal_impl<float>(h, nx, nu, N, batch, d_Ginv, d_C, d_g, d_lambda, d_z, stream);
}
gbdpcg_status gbdpcg_recover_primal_f64(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,
                                        const double *d_Ginv, const double *d_C, const double *d_g, const double *d_lambda,
                                        double *d_z, void *stream)
{
    return recover_primal_impl<double>(h, nx, nu, N, batch, d_Ginv, d_C, d_g, d_lambda, d_z, stream);
}

gbdpcg_status gbdpcg_form_pinv_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const float *d_S,
                                         float *d_Pinv, gbdpcg_pinv_kind kind, const float *d_gamma, float *d_lambda,
                                         float *d_r, float *d_p, float tol, uint32_t max_iter, uint32_t *d_iters,
                                         uint8_t *d_max_iter_exit, void *stream)
{
    return form_pinv_solve_impl<float>(h, n, N, batch, d_S, d_Pinv, kind, d_gamma, d_lambda, d_r, d_p, tol, max_iter,
                                       d_iters, d_max_iter_exit, (hipStream_t)stream);
}
gbdpcg_status gbdpcg_form_pinv_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch, const double *d_S,
                                         double *d_Pinv, gbdpcg_pinv_kind kind, const double *d_gamma, double *d_lambda,
                                         double *d_r, double *d_p, double tol, uint32_t max_iter, uint32_t *d_iters,
                                         uint8_t *d_max_iter_exit, void *stream)
{
    return form_pinv_solve_impl<double>(h, n, N, batch, d_S, d_Pinv, kind, d_gamma, d_lambda, d_r, d_p, tol, max_iter,
                                        d_iters, d_max_iter_exit, (hipStream_t)stream);
}

gbdpcg_status gbdpcg_graph_create_form_pinv_solve_f32(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                                      const float *d_S, float *d_Pinv, gbdpcg_pinv_kind kind,
                                                      const float *d_gamma, float *d_lambda, float *d_r, float *d_p,
                                                      float tol, uint32_t max_iter, uint32_t *d_iters,
                                                      uint8_t *d_max_iter_exit, gbdpcg_graph_t *out)
{
    if (!d_Pinv || (int)kind < 0 || (int)kind > 2) return GBDPCG_ERR_INVALID;
    return graph_create_impl<float>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters,
                                    d_max_iter_exit, out, (int)kind);
}
gbdpcg_status gbdpcg_graph_create_form_pinv_solve_f64(gbdpcg_handle_t h, uint32_t n, uint32_t N, uint32_t batch,
                                                      const double *d_S, double *d_Pinv, gbdpcg_pinv_kind kind,
                                                      const double *d_gamma, double *d_lambda, double *d_r, double *d_p,
                                                      double tol, uint32_t max_iter, uint32_t *d_iters,
                                                      uint8_t *d_max_iter_exit, gbdpcg_graph_t *out)
{
    if (!d_Pinv || (int)kind < 0 || (int)kind > 2) return GBDPCG_ERR_INVALID;
    return graph_create_impl<double>(h, n, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters,
                                     d_max_iter_exit, out, (int)kind);
}

#define GBDPCG_KKT_STEP(SUF, TYPE)                                                                                                  \
    gbdpcg_status gbdpcg_kkt_step_##SUF(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch, const TYPE *d_G,     \
                                        const TYPE *d_C, const TYPE *d_g, const TYPE *d_c, TYPE *d_S, TYPE *d_gamma, TYPE *d_Ginv,     \
                                        TYPE *d_Pinv, gbdpcg_pinv_kind kind, TYPE *d_lambda, TYPE *d_r, TYPE *d_p, TYPE tol,           \
                                        uint32_t max_iter, uint32_t *d_iters, uint8_t *d_max_iter_exit, TYPE *d_z, void *stream)       \
    {                                                                                                                               \
        const KktOperands<TYPE> k{nu, d_G, d_C, d_g, d_c, d_Ginv, d_z};                                                             \
        return kkt_step_impl<TYPE>(h, nx, N, batch, k, d_S, d_gamma, d_Pinv, kind, d_lambda, d_r, d_p, tol, max_iter, d_iters,       \
                                   d_max_iter_exit, (hipStream_t)stream);                                                           \
    }                                                                                                                               \
    gbdpcg_status gbdpcg_graph_create_kkt_step_##SUF(gbdpcg_handle_t h, uint32_t nx, uint32_t nu, uint32_t N, uint32_t batch,         \
                                                     const TYPE *d_G, const TYPE *d_C, const TYPE *d_g, const TYPE *d_c, TYPE *d_S,    \
                                                     TYPE *d_gamma, TYPE *d_Ginv, TYPE *d_Pinv, gbdpcg_pinv_kind kind,                \
                                                     TYPE *d_lambda, TYPE *d_r, TYPE *d_p, TYPE tol, uint32_t max_iter,               \
                                                     uint32_t *d_iters, uint8_t *d_max_iter_exit, TYPE *d_z, gbdpcg_graph_t *out)     \
    {                                                                                                                               \
        if (!d_S || !d_gamma || !d_Pinv || !d_Ginv || !d_z || (int)kind < 0 || (int)kind > 2) return GBDPCG_ERR_INVALID;           \
        const KktOperands<TYPE> k{nu, d_G, d_C, d_g, d_c, d_Ginv, d_z};                                                             \
        return graph_create_impl<TYPE>(h, nx, N, batch, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, tol, max_iter, d_iters,            \
                                       d_max_iter_exit, out, (int)kind, &k);                                                        \
    }
GBDPCG_KKT_STEP(f32, float)
GBDPCG_KKT_STEP(f64, double)
#undef GBDPCG_KKT_STEP

gbdpcg_status gbdpcg_csr_to_bt_f32(uint32_t n, uint32_t N, const uint32_t *row_ptr, const uint32_t *col_ind,
                                   const float *val, float *h_M)
{
    return csr_to_bt_impl<float>(n, N, row_ptr, col_ind, val, h_M);
}
gbdpcg_status gbdpcg_csr_to_bt_f64(uint32_t n, uint32_t N, const uint32_t *row_ptr, const uint32_t *col_ind,
                                   const double *val, double *h_M)
{
    return csr_to_bt_impl<double>(n, N, row_ptr, col_ind, val, h_M);
}

const char *gbdpcg_version(void) { return "gbdpcg 0.1 gfx950"; }

#if defined(GBDPCG_CL_STAMPS) || defined(GBDPCG_RS_STAMPS)
// diagnostic builds only (tools/cluster_stamps.py, tools/rs_stamps.py): the cluster path's workspace holds the stamps
void *gbdpcg_internal_cluster_ws(gbdpcg_handle_t h) { return h ? h->cluster_ws : nullptr; }
#endif

#ifdef GBDPCG_PERSIST_STAMPS
// diagnostic build only (tools/persist_stamps.py): where the persistent path keeps its stamps
void *gbdpcg_internal_persist_ws(gbdpcg_handle_t h) { return h ? h->pws_last : nullptr; }
#endif

}  // extern "C"
