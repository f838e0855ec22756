"""GPU edge cases of the C ABI: argument validation, unsupported shapes, misaligned pointers (narrower
vector width fallback), persistent workgroups (batch larger than the grid), long horizons that do not
fit one workgroup's LDS (automatic split path), non-default streams, occupancy check."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from gbd_pcg_amd import binding, synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    s = binding.Solver(0)
    yield s
    s.close()


def relerr(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_invalid_arguments_return_status(solver):
    lib, h = solver.lib, solver.h
    z = torch.zeros(64, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    f0 = ctypes.c_float(1e-6)
    # n == 0, N == 0, batch == 0, null S, null iters
    assert lib.gbdpcg_solve_f32(h, 0, 3, 1, p(z), None, p(z), p(z), None, None, f0, 5, p(it), None, None) == 1
    assert lib.gbdpcg_solve_f32(h, 2, 0, 1, p(z), None, p(z), p(z), None, None, f0, 5, p(it), None, None) == 1
    assert lib.gbdpcg_solve_f32(h, 2, 3, 0, p(z), None, p(z), p(z), None, None, f0, 5, p(it), None, None) == 1
    assert lib.gbdpcg_solve_f32(h, 2, 3, 1, None, None, p(z), p(z), None, None, f0, 5, p(it), None, None) == 1
    assert lib.gbdpcg_solve_f32(h, 2, 3, 1, p(z), None, p(z), p(z), None, None, f0, 5, None, None, None) == 1
    assert lib.gbdpcg_spmv_f32(h, 2, 3, 1, p(z), None, p(z), None) == 1
    # a block size no lane map covers: odd n > 64 (n/V > 64 for every V)
    assert lib.gbdpcg_spmv_f32(h, 67, 2, 1, p(z), p(z), p(z), None) == 4      # GBDPCG_ERR_UNSUPPORTED
    assert lib.gbdpcg_check_occupancy(h, 4, 67, 2, 1) == 4
    assert lib.gbdpcg_check_occupancy(h, 4, 14, 128, 1024) == 0
    assert lib.gbdpcg_check_occupancy(h, 8, 36, 256, 1) == 0                  # split path, fits
    assert lib.gbdpcg_check_occupancy(h, 3, 14, 8, 1) == 1                    # elem_size must be 4 or 8


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_misaligned_matrix_pointers_use_narrower_loads(solver, orc, dtype):
    """S / Pinv offset by one element: only V = 1 loads are legal; results must not change."""
    n, N, B = 14, 12, 2
    d = synth.gen_numpy(n, N, seed=91, batch=B, dtype=dtype)
    es = np.dtype(dtype).itemsize
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    bufS = torch.zeros(d["S"].size + 1, dtype=tdt, device="cuda")
    bufP = torch.zeros(d["Pinv"].size + 1, dtype=tdt, device="cuda")
    S, P = bufS[1:], bufP[1:]
    assert S.data_ptr() % (2 * es) != 0
    S.copy_(dev(d["S"]).reshape(-1))
    P.copy_(dev(d["Pinv"]).reshape(-1))
    g = dev(d["gamma"])
    for path in (binding.PATH_FUSED, binding.PATH_SPLIT):
        solver.set_path(path)
        lam = torch.zeros_like(g)
        iters, flags = solver.solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=50)
        torch.cuda.synchronize()
        ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=50)
        assert np.array_equal(iters.cpu().numpy(), ob["iters"].astype(np.int32))
        lam_h = lam.cpu().numpy().reshape(B, -1)
        for b in range(B):
            assert relerr(lam_h[b], ob["lambda_"][b]) < (1e-10 if dtype == np.float64 else 1e-6)
    solver.set_path(binding.PATH_AUTO)
    x = torch.randn(B * n * N, dtype=tdt, device="cuda")
    y = solver.spmv(n, N, B, S, x)
    y_ref = solver.spmv(n, N, B, dev(d["S"]).reshape(-1), x)
    torch.cuda.synchronize()
    assert relerr(y.cpu().numpy(), y_ref.cpu().numpy()) < 1e-6


def test_batch_larger_than_grid(solver, orc):
    """More problems than resident workgroups: the fused kernel's persistent loop must visit all."""
    n, N, B = 6, 5, 3000
    base = synth.gen_numpy(n, N, seed=17, batch=8, dtype=np.float64)
    idx = np.arange(B) % 8
    S, P, g = base["S"][idx], base["Pinv"][idx], base["gamma"][idx] * (1.0 + 0.001 * np.arange(B))[:, None]
    solver.set_path(binding.PATH_FUSED)
    lam = torch.zeros((B, n * N), dtype=torch.float64, device="cuda")
    iters, flags = solver.solve(n, N, B, dev(S), dev(P), dev(g), lam, tol=1e-22, max_iter=60)
    torch.cuda.synchronize()
    solver.set_path(binding.PATH_AUTO)
    assert flags.sum().item() == 0
    lam = lam.cpu().numpy()
    for b in (0, 1, 7, 8, 1023, 1024, 2047, 2999):
        A = orc.dense_from_bt(n, N, S[b])
        assert relerr(lam[b], np.linalg.solve(A, g[b])) < 1e-8


def test_long_horizon_goes_split_automatically(solver, orc):
    """n=14, N=1200 fp64: four vectors of 134 KB exceed one workgroup's LDS -> AUTO picks the split path."""
    n, N = 14, 1200
    assert solver.choose_path(8, n, N, 1) == binding.PATH_SPLIT
    d = synth.gen_numpy(n, N, seed=23, dtype=np.float64)
    lam = torch.zeros(n * N, dtype=torch.float64, device="cuda")
    iters, flags = solver.solve(n, N, 1, dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"]), lam, tol=1e-6, max_iter=60)
    torch.cuda.synchronize()
    o = orc.pcg(n, N, d["S"][0], d["Pinv"][0], d["gamma"][0], tol=1e-6, max_iter=60)
    assert int(iters[0]) == o["iters"] and relerr(lam.cpu().numpy(), o["lambda_"]) < 1e-10


@pytest.mark.parametrize("path", [binding.PATH_FUSED, binding.PATH_SPLIT])
def test_non_default_stream(solver, orc, path):
    n, N, B = 14, 10, 5
    d = synth.gen_numpy(n, N, seed=29, batch=B, dtype=np.float32)
    s = torch.cuda.Stream()
    S, P, g = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
    lam = torch.zeros_like(g)
    torch.cuda.synchronize()
    solver.set_path(path)
    with torch.cuda.stream(s):
        iters, flags = solver.solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=50, stream=s)
    s.synchronize()
    solver.set_path(binding.PATH_AUTO)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=50)
    assert np.array_equal(iters.cpu().numpy(), ob["iters"].astype(np.int32))
    for b in range(B):
        assert relerr(lam.cpu().numpy()[b], ob["lambda_"][b]) < 1e-6


def test_register_resident_path_limits(solver, orc):
    """n=14 fp32: N <= 72 runs register-resident, N = 73 streams; both must agree with the oracle
    (N = 63, 64, 72, 73 straddle the wave / workgroup boundaries of the resident lane map)."""
    n = 14
    for N in (1, 2, 9, 10, 63, 64, 72, 73):
        d = synth.gen_numpy(n, N, seed=400 + N, batch=2, dtype=np.float32)
        lam = torch.zeros((2, n * N), dtype=torch.float32, device="cuda")
        iters, flags = solver.solve(n, N, 2, dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"]), lam, tol=1e-6, max_iter=60)
        torch.cuda.synchronize()
        ob = orc.pcg_batch(n, N, 2, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=60)
        assert np.array_equal(iters.cpu().numpy(), ob["iters"].astype(np.int32)), N
        for b in range(2):
            assert relerr(lam.cpu().numpy()[b], ob["lambda_"][b]) < 1e-6, N


def test_back_to_back_graph_replays_keep_the_symmetric_path(solver):
    """Replaying the default-mode graph many times without host synchronisation, with other kernels
    interleaved on the legacy stream, must not change which kernel solves the problems: the symmetry
    flags are initialised by a kernel node (a hipGraph memset node wrote garbage in exactly this pattern
    and every problem silently fell back to the general streaming kernel, 4x slower, same answers)."""
    n, N, B = 14, 128, 512
    g = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=77)
    S, gamma = g["S"], g["gamma"]
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    graph = solver.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0, 10, it, fl)

    def per_replay(sync):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(20):
            lam.zero_()
            graph.launch()
            if sync:
                torch.cuda.synchronize()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / 20

    per_replay(True)
    synced = per_replay(True)
    lam_synced = lam.clone()
    queued = per_replay(False)
    graph.close()
    assert int(it.min()) == 10 and int(it.max()) == 10
    # the kernels differ in summation order: had any problem of a queued replay gone to another kernel, its lambda would
    # differ in the last bits from the synchronised replays' (since round 2 the general path is only 1.4x slower than the
    # symmetric one, so time alone no longer tells)
    assert torch.equal(lam, lam_synced)
    assert queued < 2.0 * synced, (queued, synced)


def test_solve_inside_a_caller_owned_capture(solver, orc):
    """gbdpcg_solve_* is capture-safe once gbdpcg_reserve has been called for the shape: a solve captured into a
    graph the CALLER owns (here torch's) -- default symmetric mode, so the device check, the resident symmetric
    launch and the general launch are all inside -- replays to the oracle's answer."""
    n, N, B = 14, 90, 5
    d = synth.gen_numpy(n, N, seed=404, batch=B, dtype=np.float32)
    S, g = dev(d["S"]), dev(d["gamma"])
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(g)
    iters = torch.zeros(B, dtype=torch.int32, device="cuda")
    flags = torch.zeros(B, dtype=torch.uint8, device="cuda")
    solver.reserve(4, n, N, B)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        solver.solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=40, iters=iters, max_iter_exit=flags)
    for _ in range(2):
        lam.zero_()
        graph.replay()
    torch.cuda.synchronize()
    ob = orc.pcg_batch(n, N, B, d["S"], P.cpu().numpy(), d["gamma"], tol=1e-6, max_iter=40)
    assert np.array_equal(iters.cpu().numpy().astype(np.int64), ob["iters"].astype(np.int64))
    lam_h = lam.cpu().numpy().reshape(B, -1)
    for b in range(B):
        assert np.linalg.norm(lam_h[b] - ob["lambda_"][b]) / np.linalg.norm(ob["lambda_"][b]) < 1e-6


def test_sliced_persistent_solve_inside_a_caller_owned_capture(orc):
    """The same for a small batch of large problems, which AUTO cuts into persistent launches in a row (api.hip, persist_slices):
    gbdpcg_reserve prepares the hand-off words of the slices (and the split path's workspace -- which of the two a solve takes
    depends on its max_iter), so both captures below are legal; a fresh handle, so that nothing is there by accident."""
    n, N, B = 24, 100, 7
    s = binding.Solver(0)
    try:
        d = synth.gen_numpy(n, N, seed=405, batch=B, dtype=np.float32)
        S, P, g = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
        iters = torch.zeros(B, dtype=torch.int32, device="cuda")
        flags = torch.zeros(B, dtype=torch.uint8, device="cuda")
        s.reserve(4, n, N, B)
        torch.cuda.synchronize()
        for max_iter in (60, 8):   # 60: persistent launches in a row; 8: the split graph is the shorter one
            lam = torch.zeros_like(g)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                s.solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=max_iter, iters=iters, max_iter_exit=flags)
            for _ in range(2):
                lam.zero_()
                graph.replay()
            torch.cuda.synchronize()
            ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=max_iter, nthreads=4)
            assert np.array_equal(iters.cpu().numpy().astype(np.int64), ob["iters"].astype(np.int64)), max_iter
            assert np.array_equal(flags.cpu().numpy().astype(bool), ob["max_iter_exit"].astype(bool))
            lam_h = lam.cpu().numpy().reshape(B, -1)
            for b in range(B):
                assert np.linalg.norm(lam_h[b] - ob["lambda_"][b]) / np.linalg.norm(ob["lambda_"][b]) < 2e-6, (max_iter, b)
            del graph
    finally:
        s.close()


@pytest.mark.parametrize("shape", [(14, 40, 3, 400, np.float32), (36, 256, 1, 4, np.float64)])
def test_graph_survives_growth_of_the_handle_buffers(orc, shape):
    """A graph holds the handle's scratch (the verdict bytes of the device symmetry check; the split path's
    workspace) in its kernel nodes.  A later, larger solve through the same handle must grow that scratch without
    freeing what the graph points at: replay the first graph afterwards and compare with the oracle."""
    n, N, B, B_big, dtype = shape
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    tol = 1e-6 if dtype == np.float32 else 1e-10
    s = binding.Solver(0)  # a fresh handle: nothing reserved yet
    try:
        d = synth.gen_numpy(n, N, seed=515, batch=B, dtype=dtype)
        S, g = dev(d["S"]), dev(d["gamma"])
        P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
        lam = torch.zeros_like(g)
        iters = torch.zeros(B, dtype=torch.int32, device="cuda")
        flags = torch.zeros(B, dtype=torch.uint8, device="cuda")
        graph = s.graph_solve(n, N, B, S, P, g, lam, None, None, 1e-6, 40, iters, flags)
        graph.launch()
        torch.cuda.synchronize()
        first = lam.clone()
        # larger batch through the same handle: forces both scratch buffers to grow
        big = synth.gen_torch(n, N, B_big, "cuda", tdt, seed=99)
        Pb = s.form_pinv(n, N, B_big, big["S"], binding.PINV_STAIR)
        lb = torch.zeros_like(big["gamma"])
        s.solve(n, N, B_big, big["S"], Pb, big["gamma"], lb, tol=1e-6, max_iter=40)
        torch.cuda.synchronize()
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]  # reuse of any freed block
        lam.zero_()
        graph.launch()
        torch.cuda.synchronize()
        del junk
        graph.close()
        assert torch.equal(lam, first)
        ob = orc.pcg_batch(n, N, B, d["S"], P.cpu().numpy(), d["gamma"], tol=1e-6, max_iter=40)
        assert np.array_equal(iters.cpu().numpy().astype(np.int64), ob["iters"].astype(np.int64))
        lam_h = lam.cpu().numpy().reshape(B, -1)
        for b in range(B):
            assert relerr(lam_h[b], ob["lambda_"][b]) < tol
    finally:
        s.close()


def _two_calls(solver, n, N, B, S, g, kind, tol, max_iter):
    P = solver.form_pinv(n, N, B, S, kind)
    lam = torch.zeros_like(g)
    it, fl = solver.solve(n, N, B, S, P, g, lam, tol=tol, max_iter=max_iter)
    torch.cuda.synchronize()
    return P, lam, it, fl


@pytest.mark.parametrize("shape", [(14, 128, 7, np.float32), (14, 31, 3, np.float32), (14, 16, 2, np.float32),
                                   (12, 50, 4, np.float32), (6, 40, 3, np.float64), (36, 20, 2, np.float64),
                                   (5, 9, 2, np.float32), (14, 1, 2, np.float32)])
@pytest.mark.parametrize("kind", [binding.PINV_STAIR, binding.PINV_BLOCK_JACOBI, binding.PINV_IDENTITY])
def test_form_pinv_solve_equals_the_two_calls(solver, shape, kind):
    """gbdpcg_form_pinv_solve_* = gbdpcg_form_pinv_* then gbdpcg_solve_*, bit for bit: the stair kernel's symmetry
    verdicts (where it reports them) select the same kernels as the solve's own test would.  Shapes with and
    without the one-launch stair kernel, with and without a symmetric solve kernel, N = 1."""
    n, N, B, dtype = shape
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    d = synth.gen_torch(n, N, B, "cuda", tdt, seed=2024)
    S, g = d["S"], d["gamma"]
    P_ref, lam_ref, it_ref, fl_ref = _two_calls(solver, n, N, B, S, g, kind, 1e-6, 30)
    P = torch.full_like(S, float("nan"))
    lam = torch.zeros_like(g)
    it, fl = solver.form_pinv_solve(n, N, B, S, P, g, lam, kind=kind, tol=1e-6, max_iter=30)
    torch.cuda.synchronize()
    assert torch.equal(P, P_ref)
    assert torch.equal(it, it_ref) and torch.equal(fl, fl_ref)
    assert torch.equal(lam, lam_ref)


def test_form_pinv_solve_with_some_asymmetric_problems(solver, orc):
    """Problems whose S has one L_{k+1} != R_k^T must be solved by the general kernel (their Pinv pair is formed from
    both sides); the others by the resident symmetric one; all must match the oracle run on the same S and Pinv."""
    n, N, base = 14, 100, 6
    # 132 problems: more than one round of two-workgroup clusters holds (128), so that mode 2 does run its two-kernel
    # dispatch (smaller batches go to the cluster kernel as a whole: api.hip, one_cluster_round)
    B = 132
    d0 = synth.gen_numpy(n, N, seed=31337, batch=base, dtype=np.float32)
    idx = np.arange(B) % base
    d = {"S": d0["S"][idx].copy(), "gamma": (d0["gamma"][idx] * (1.0 + 0.001 * (np.arange(B) // base))[:, None]).astype(np.float32)}
    S_h = d["S"].reshape(B, N, 3, n, n).copy()
    S_h[1, 40, 0, 3, 5] *= 1.0 + 2.0 ** -20   # L_40 of problem 1: last-bits perturbation
    S_h[4, N - 1, 0, 0, 0] += 1e-3            # L_{N-1} of problem 4
    S = dev(S_h.reshape(-1))
    g = dev(d["gamma"])
    P = torch.empty_like(S)
    lam = torch.zeros_like(g)
    it, fl = solver.form_pinv_solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=60)
    torch.cuda.synchronize()
    flags = solver.check_symmetric(n, N, B, P).cpu().numpy()
    assert list(flags[:6]) == [1, 0, 1, 1, 0, 1] and flags[6:].all()   # Pinv is exactly symmetric exactly where S was
    P_ref, lam_ref, it_ref, _ = _two_calls(solver, n, N, B, S, g, binding.PINV_STAIR, 1e-6, 60)
    assert torch.equal(P, P_ref) and torch.equal(lam, lam_ref) and torch.equal(it, it_ref)
    ob = orc.pcg_batch(n, N, B, S_h.reshape(-1), P.cpu().numpy(), d["gamma"], tol=1e-6, max_iter=60, nthreads=8)
    # a last-bits asymmetry leaves the operator (numerically) what it was; the iteration counts agree with the oracle's
    assert np.array_equal(it.cpu().numpy().astype(np.int64), ob["iters"].astype(np.int64))
    lam_h = lam.cpu().numpy().reshape(B, -1)
    for b in range(B):
        assert relerr(lam_h[b], ob["lambda_"][b]) < 2e-6


def test_form_pinv_solve_graph_replays_and_skips_the_test_launch(solver):
    """The captured form + solve replays to the same answer after S is rewritten in place, and costs less than the
    captured solve alone plus a separate Pinv formation (no symmetry-test launch: 70 us of ~0.5 ms here)."""
    n, N, B = 14, 128, 1024
    d = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=5)
    S, g = d["S"], d["gamma"]
    P = torch.empty_like(S)
    lam = torch.zeros_like(g)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    both = solver.graph_form_pinv_solve(n, N, B, S, P, g, lam, None, None, 1e-6, 40, it, fl)
    both.launch()
    torch.cuda.synchronize()
    P_ref, lam_ref, it_ref, _ = _two_calls(solver, n, N, B, S, g, binding.PINV_STAIR, 1e-6, 40)
    assert torch.equal(P, P_ref) and torch.equal(lam, lam_ref) and torch.equal(it.to(it_ref.dtype), it_ref)
    # new S in the same buffers
    d2 = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=6)
    S.copy_(d2["S"]); g.copy_(d2["gamma"]); lam.zero_()
    both.launch()
    torch.cuda.synchronize()
    P_ref, lam_ref, it_ref, _ = _two_calls(solver, n, N, B, S, g, binding.PINV_STAIR, 1e-6, 40)
    assert torch.equal(P, P_ref) and torch.equal(lam, lam_ref) and torch.equal(it.to(it_ref.dtype), it_ref)

    solve_only = solver.graph_solve(n, N, B, S, P, g, lam, None, None, 1e-6, 40, it, fl)

    def timed(fn):
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(20):
            fn()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / 20

    def fused():
        lam.zero_(); both.launch()

    def separate():
        lam.zero_(); solver.form_pinv(n, N, B, S, binding.PINV_STAIR, Pinv=P); solve_only.launch()

    t_fused, t_sep = timed(fused), timed(separate)
    both.close(); solve_only.close()
    assert t_fused < t_sep, (t_fused, t_sep)   # ms; the difference is ~0.08 ms, the bound leaves room for a noisy box
