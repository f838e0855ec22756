"""The cluster path (csrc/pcg_cluster.hip): general-storage problems of stateSize 14, fp32, 72 < knotPoints <= 288 with both
matrices register-resident over 2-4 compute units, against the CPU oracle, through the C ABI.  It is what the fused path
runs when the symmetric kernels do not apply (symmetric mode 0, or storage that fails the bit-for-bit test).  Tolerances as
in test_gpu_parity.py: fp32 1e-6 norm-wise, equal iteration counts on the a = 0.5 generator."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from gbd_pcg_amd import binding, synth  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def solver():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    s = binding.Solver(0)
    yield s
    s.close()


def relerr(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unsymmetrize(M, n, N, B, every=1):
    """Make the storage of every `every`-th problem fail the bit-for-bit symmetry test without changing the matrix much:
    one element of one L block is moved by one ulp."""
    M = M.reshape(B, N, 3, n, n).copy()
    for b in range(0, B, every):
        k = 1 + (7 * b) % (N - 1)
        M[b, k, 0, 3, 5] = np.nextafter(M[b, k, 0, 3, 5], np.float32(np.inf))
    return M.reshape(B, -1)


def run(solver, n, N, B, S, Pinv, gamma, lam0=None, tol=1e-6, max_iter=100, symmetric=0):
    es = np.dtype(S.dtype).itemsize
    assert solver.cluster_members(es, n, N) >= 1, "shape has no cluster form"
    assert solver.choose_path(es, n, N, B) == binding.PATH_FUSED
    solver.set_symmetric(symmetric)
    try:
        dS, dP, dg = dev(S), dev(Pinv), dev(gamma)
        lam = torch.zeros_like(dg) if lam0 is None else dev(lam0)
        r, p = torch.full_like(dg, float("nan")), torch.full_like(dg, float("nan"))
        it, fl = solver.solve(n, N, B, dS, dP, dg, lam, r, p, tol=tol, max_iter=max_iter)
        torch.cuda.synchronize()
    finally:
        solver.set_symmetric(2)
    return dict(lambda_=lam.cpu().numpy().reshape(B, -1), r=r.cpu().numpy().reshape(B, -1),
                p=p.cpu().numpy().reshape(B, -1), iters=it.cpu().numpy().astype(np.int64),
                flag=fl.cpu().numpy().astype(np.int64))


def check(out, ob, d, B, ltol=1e-6, vtol=2e-5):
    assert np.array_equal(out["iters"], ob["iters"].astype(np.int64)), (out["iters"], ob["iters"])
    assert np.array_equal(out["flag"], ob["max_iter_exit"].astype(np.int64))
    for b in range(B):
        assert relerr(out["lambda_"][b], ob["lambda_"][b]) < ltol, b
        scale = np.abs(d["gamma"][b]).max()
        assert np.abs(out["r"][b] - ob["r"].reshape(B, -1)[b]).max() < vtol * scale, b
        assert np.abs(out["p"][b] - ob["p"].reshape(B, -1)[b]).max() < vtol * scale, b


def test_cluster_shapes(solver):
    """Members per problem: 0 = no cluster form (not built for the block size, or pcg_resident.hip has the problem in one workgroup);
    1 = a "cluster" of one workgroup where pcg_resident.hip is not built for the block size (16, 18, fp64 from 14 on)."""
    assert solver.cluster_members(4, 14, 128) == 2 and solver.cluster_members(4, 14, 144) == 2
    assert solver.cluster_members(4, 14, 145) == 3 and solver.cluster_members(4, 14, 288) == 4
    assert solver.cluster_members(4, 14, 72) == 0      # one workgroup holds it (pcg_resident.hip)
    assert solver.cluster_members(4, 14, 289) == 5 and solver.cluster_members(4, 14, 576) == 8 and solver.cluster_members(4, 14, 577) == 0
    assert solver.cluster_members(4, 36, 128) == 0
    # stateSize 12 (round 3): 80 knots per workgroup
    assert solver.cluster_members(4, 12, 80) == 0 and solver.cluster_members(4, 12, 81) == 2 and solver.cluster_members(4, 12, 128) == 2
    assert solver.cluster_members(4, 12, 161) == 3 and solver.cluster_members(4, 12, 320) == 4 and solver.cluster_members(4, 12, 321) == 5 and solver.cluster_members(4, 12, 641) == 0
    # one row per lane: stateSize 13 in fp32 and the BASELINE block size in fp64 (32 knots per workgroup)
    assert solver.cluster_members(4, 13, 32) == 0 and solver.cluster_members(4, 13, 33) == 2 and solver.cluster_members(4, 13, 128) == 4
    assert solver.cluster_members(8, 14, 32) == 1 and solver.cluster_members(8, 14, 64) == 2 and solver.cluster_members(8, 14, 128) == 4
    assert solver.cluster_members(8, 14, 129) == 0 and solver.cluster_members(8, 36, 64) == 0
    # ... 8, 10 and 16: 128, 96 and 64 knots per workgroup
    assert solver.cluster_members(4, 8, 128) == 0 and solver.cluster_members(4, 8, 256) == 2 and solver.cluster_members(4, 8, 512) == 4
    assert solver.cluster_members(4, 10, 96) == 0 and solver.cluster_members(4, 10, 128) == 2
    assert solver.cluster_members(4, 16, 64) == 1 and solver.cluster_members(4, 16, 128) == 2 and solver.cluster_members(4, 16, 256) == 4
    # one row per lane at the other block sizes: fp64 8 / 10 / 12 (64 / 48 / 40 knots per workgroup), fp32 9 / 11 / 15 (56 / 40 / 32)
    assert solver.cluster_members(8, 12, 40) == 0 and solver.cluster_members(8, 12, 128) == 4 and solver.cluster_members(8, 12, 161) == 0
    assert solver.cluster_members(8, 10, 48) == 0 and solver.cluster_members(8, 10, 128) == 3 and solver.cluster_members(8, 8, 128) == 2
    assert solver.cluster_members(4, 9, 56) == 0 and solver.cluster_members(4, 9, 128) == 3 and solver.cluster_members(4, 11, 128) == 4
    assert solver.cluster_members(4, 15, 32) == 0 and solver.cluster_members(4, 15, 128) == 4 and solver.cluster_members(4, 15, 129) == 5 and solver.cluster_members(4, 15, 257) == 0
    # stateSize 18: 9 lanes per knot, 56 knots per workgroup, the D and R blocks of Pinv in LDS
    assert solver.cluster_members(4, 18, 56) == 1 and solver.cluster_members(4, 18, 128) == 3 and solver.cluster_members(4, 18, 224) == 4
    assert solver.cluster_members(4, 18, 225) == 5 and solver.cluster_members(4, 18, 449) == 0 and solver.cluster_members(8, 18, 128) == 0 and solver.cluster_members(4, 20, 128) == 0
    # the small blocks beyond pcg_resident.hip's horizons (512 / 256 / 168 knots in fp32); the reference's example system (2 x 3) streams
    assert solver.cluster_members(4, 2, 512) == 0 and solver.cluster_members(4, 2, 513) == 2 and solver.cluster_members(4, 2, 3) == 0
    assert solver.cluster_members(4, 4, 1024) == 4 and solver.cluster_members(4, 4, 1025) == 5 and solver.cluster_members(8, 6, 81) == 2
    # fp64 at stateSize 16: 16 lanes per knot, 32 knots per workgroup, the D and R blocks of Pinv in LDS
    assert solver.cluster_members(8, 16, 32) == 1 and solver.cluster_members(8, 16, 128) == 4 and solver.cluster_members(8, 16, 129) == 0


@pytest.mark.parametrize("N,B", [(128, 5), (127, 3), (73, 2), (100, 9), (144, 3), (145, 2), (200, 4), (216, 1), (217, 2), (288, 3),
                                 (289, 2), (300, 40), (433, 2), (505, 1), (576, 3)])   # five to eight members (fp32 only)
def test_cluster_vs_oracle(solver, orc, N, B):
    """Two, three and four workgroups per problem, even and ragged splits, more and fewer problems than one round."""
    n = 14
    d = synth.gen_numpy(n, N, seed=500 + N, batch=B, dtype=np.float32)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100)
    out = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"])
    check(out, ob, d, B)


@pytest.mark.parametrize("n,N,B", [(12, 128, 5), (12, 81, 3), (12, 100, 140), (12, 160, 2), (12, 161, 3), (12, 240, 70), (12, 320, 2),
                                   (12, 319, 1), (8, 129, 3), (8, 256, 70), (8, 300, 2), (8, 512, 1), (10, 97, 2), (10, 128, 140), (10, 200, 3),
                                   (10, 384, 2), (16, 65, 2), (16, 128, 140), (16, 129, 3), (16, 200, 70), (16, 256, 2),
                                   (18, 57, 2), (18, 112, 3), (18, 113, 2), (18, 128, 140), (18, 168, 70), (18, 224, 2),
                                   (12, 400, 3), (12, 640, 1), (8, 1000, 2), (10, 700, 2), (16, 300, 40), (16, 512, 1), (18, 300, 2), (18, 448, 1)])
def test_cluster_other_state_sizes(solver, orc, n, N, B):
    """The same kernel at stateSize 8, 10, 12, 16 and 18 (n / 2 lanes per knot, 128 / 96 / 80 / 64 / 56 knots per workgroup; at 16 the R
    block of Pinv stays in LDS, at 18 its D block as well): two, three and four workgroups per problem, ragged splits, more problems
    than one round of clusters -- in general storage and, since symmetric STREAMING is slower than general RESIDENT, in the
    default symmetric mode as well."""
    base = min(B, 6)
    d = synth.gen_numpy(n, N, seed=700 + N, batch=base, dtype=np.float32)
    idx = np.arange(B) % base
    S, Pi = d["S"][idx], d["Pinv"][idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // base))[:, None]).astype(np.float32)
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=1e-6, max_iter=100, nthreads=8)
    for mode in (0, 2):
        out = run(solver, n, N, B, S, Pi, g, symmetric=mode)
        check(out, ob, {"gamma": g}, B)
    # fixed count, warm start: lambda, r and p after exactly 5 iterations
    lam0 = (0.1 * np.random.default_rng(N).standard_normal((B, n * N))).astype(np.float32)
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=0.0, max_iter=5, lambda0=lam0, nthreads=8)
    out = run(solver, n, N, B, S, Pi, g, lam0=lam0, tol=0.0, max_iter=5)
    check(out, ob, {"gamma": g}, B, ltol=2e-6)


@pytest.mark.parametrize("n,dtype,N,B", [(14, np.float64, 128, 5), (14, np.float64, 33, 3), (14, np.float64, 64, 140), (14, np.float64, 65, 2),
                                         (14, np.float64, 100, 70), (14, np.float64, 127, 1), (13, np.float32, 128, 5), (13, np.float32, 33, 3),
                                         (13, np.float32, 64, 140), (13, np.float32, 97, 70), (13, np.float32, 100, 2),
                                         (12, np.float64, 128, 5), (12, np.float64, 41, 3), (12, np.float64, 160, 2), (12, np.float64, 100, 70),
                                         (10, np.float64, 49, 2), (10, np.float64, 128, 70), (10, np.float64, 192, 2), (8, np.float64, 65, 3),
                                         (8, np.float64, 200, 70), (8, np.float64, 256, 2), (9, np.float32, 57, 3), (9, np.float32, 128, 70),
                                         (9, np.float32, 224, 2), (11, np.float32, 41, 2), (11, np.float32, 128, 70), (11, np.float32, 160, 3),
                                         (15, np.float32, 33, 3), (15, np.float32, 100, 70), (15, np.float32, 128, 5),
                                         (16, np.float64, 33, 3), (16, np.float64, 100, 2), (16, np.float64, 128, 70),
                                         (3, np.float32, 169, 3), (3, np.float32, 500, 70), (5, np.float32, 128, 70), (5, np.float32, 384, 2),
                                         (7, np.float32, 73, 3), (7, np.float32, 128, 70), (7, np.float32, 288, 2), (3, np.float64, 300, 5),
                                         (5, np.float64, 97, 70), (7, np.float64, 128, 5), (9, np.float64, 57, 3), (9, np.float64, 224, 2),
                                         (11, np.float64, 128, 70), (13, np.float64, 33, 3), (13, np.float64, 128, 5), (15, np.float64, 128, 70),
                                         (2, np.float32, 513, 3), (2, np.float32, 2048, 2), (2, np.float32, 700, 70), (4, np.float32, 257, 3),
                                         (4, np.float32, 600, 70), (4, np.float32, 1024, 1), (6, np.float32, 169, 70), (6, np.float32, 672, 2),
                                         (2, np.float64, 300, 70), (2, np.float64, 1024, 2), (4, np.float64, 129, 3), (4, np.float64, 512, 5),
                                         (6, np.float64, 81, 70), (6, np.float64, 320, 2), (13, np.float32, 200, 3), (13, np.float32, 256, 1), (9, np.float32, 400, 40),
                                         (7, np.float32, 500, 70), (3, np.float32, 1300, 2), (15, np.float32, 250, 2), (11, np.float32, 300, 3)])
def test_cluster_one_row_per_lane(solver, orc, n, dtype, N, B):
    """One row per lane: the BASELINE block size in fp64 (14 lanes per knot, 32 knots per workgroup, the
    hand-off words carry both halves of an fp64 value under their own tags) and stateSize 13 in fp32 (13 lanes per knot, direct
    tile loads, one accumulator chain per row); the same template at stateSize 8, 10, 12, 16 in fp64 (at 16 the D and R blocks of Pinv live in LDS), 9, 11, 15 in fp32, and at the small sizes (2 ... 7; in fp64 every odd size up to 15) for horizons beyond pcg_resident.hip's.  fp64 to 1e-10, fp32 to 1e-6, equal iteration counts; in general storage and in
    the default symmetric mode (general RESIDENT beats symmetric STREAMING); then a fixed count from a warm start."""
    base = min(B, 6)
    d = synth.gen_numpy(n, N, seed=300 + N + n, batch=base, dtype=dtype)
    idx = np.arange(B) % base
    S, Pi = d["S"][idx], d["Pinv"][idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // base))[:, None]).astype(dtype)
    ltol = 1e-10 if dtype == np.float64 else 1e-6
    vtol = 1e-9 if dtype == np.float64 else 2e-5
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=1e-6, max_iter=100, nthreads=8)
    for mode in (0, 2):
        out = run(solver, n, N, B, S, Pi, g, symmetric=mode)
        check(out, ob, {"gamma": g}, B, ltol=ltol, vtol=vtol)
    lam0 = (0.1 * np.random.default_rng(N).standard_normal((B, n * N))).astype(dtype)
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=0.0, max_iter=5, lambda0=lam0, nthreads=8)
    out = run(solver, n, N, B, S, Pi, g, lam0=lam0, tol=0.0, max_iter=5)
    check(out, ob, {"gamma": g}, B, ltol=2 * ltol, vtol=vtol)
    # without a preconditioner, fixed count
    ob = orc.pcg_batch(n, N, B, S, None, g, tol=0.0, max_iter=4, lambda0=lam0, nthreads=8)
    out = run(solver, n, N, B, S, None, g, lam0=lam0, tol=0.0, max_iter=4)
    check(out, ob, {"gamma": g}, B, ltol=20 * ltol, vtol=10 * vtol)


@pytest.mark.parametrize("n,dtype,N,B", [(16, np.float32, 10, 3), (16, np.float32, 64, 300), (16, np.float32, 33, 1), (18, np.float32, 56, 70),
                                         (18, np.float32, 20, 2), (14, np.float64, 32, 300), (14, np.float64, 5, 2), (16, np.float64, 32, 70),
                                         (16, np.float64, 9, 1), (15, np.float64, 32, 5), (15, np.float64, 3, 70)])
def test_cluster_of_one(solver, orc, n, dtype, N, B):
    """Horizons one workgroup holds, at the block sizes pcg_resident.hip is not built for: the cluster kernel with a single member
    (no hand-off: the wave partials meet in LDS).  To tolerance with equal iteration counts, a fixed count from a warm start,
    and without a preconditioner; in general storage and in the default symmetric mode."""
    es = np.dtype(dtype).itemsize
    assert solver.cluster_members(es, n, N) == 1
    base = min(B, 6)
    d = synth.gen_numpy(n, N, seed=1300 + 7 * N + n, batch=base, dtype=dtype)
    idx = np.arange(B) % base
    S, Pi = d["S"][idx], d["Pinv"][idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // base))[:, None]).astype(dtype)
    ltol = 1e-10 if dtype == np.float64 else 1e-6
    vtol = 1e-9 if dtype == np.float64 else 2e-5
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=1e-6, max_iter=100, nthreads=8)
    for mode in (0, 2):
        out = run(solver, n, N, B, S, Pi, g, symmetric=mode)
        check(out, ob, {"gamma": g}, B, ltol=ltol, vtol=vtol)
    lam0 = (0.1 * np.random.default_rng(N).standard_normal((B, n * N))).astype(dtype)
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=0.0, max_iter=5, lambda0=lam0, nthreads=8)
    out = run(solver, n, N, B, S, Pi, g, lam0=lam0, tol=0.0, max_iter=5)
    check(out, ob, {"gamma": g}, B, ltol=2 * ltol, vtol=vtol)
    ob = orc.pcg_batch(n, N, B, S, None, g, tol=0.0, max_iter=4, lambda0=lam0, nthreads=8)
    out = run(solver, n, N, B, S, None, g, lam0=lam0, tol=0.0, max_iter=4)
    check(out, ob, {"gamma": g}, B, ltol=20 * ltol, vtol=10 * vtol)


@pytest.mark.parametrize("n,dtype,N", [(14, np.float32, 128), (16, np.float32, 128), (18, np.float32, 128), (12, np.float32, 100),
                                       (14, np.float64, 64), (16, np.float64, 100), (13, np.float32, 100)])
def test_cluster_matrices_aligned_to_8_bytes_only(solver, orc, n, dtype, N):
    """Matrices that start 8 bytes past a 16-byte boundary: the coalesced LDS-DMA tile stages need 16-byte alignment, so the
    kernel takes its direct tile loads (pcg_cluster_kernel<..., false>) -- at stateSize 16 / 18 and in fp64 the columns of Pinv
    that live in LDS then go through the registers once per problem.  Same results as from aligned buffers."""
    B = 3
    es = np.dtype(dtype).itemsize
    shift = 8 // es
    d = synth.gen_numpy(n, N, seed=900 + n, batch=B, dtype=dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    bufS = torch.zeros(d["S"].size + 4, dtype=tdt, device="cuda")
    bufP = torch.zeros(d["Pinv"].size + 4, dtype=tdt, device="cuda")
    assert bufS.data_ptr() % 16 == 0 and bufP.data_ptr() % 16 == 0
    dS, dP = bufS[shift:shift + d["S"].size], bufP[shift:shift + d["Pinv"].size]
    assert dS.data_ptr() % 16 == 8 and dP.data_ptr() % 16 == 8
    dS.copy_(dev(d["S"]).reshape(-1))
    dP.copy_(dev(d["Pinv"]).reshape(-1))
    dg = dev(d["gamma"])
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100)
    assert solver.cluster_members(es, n, N) >= 2
    solver.set_symmetric(0)
    try:
        lam = torch.zeros_like(dg)
        r, p = torch.full_like(dg, float("nan")), torch.full_like(dg, float("nan"))
        it, fl = solver.solve(n, N, B, dS, dP, dg, lam, r, p, tol=1e-6, max_iter=100)
        torch.cuda.synchronize()
    finally:
        solver.set_symmetric(2)
    out = dict(lambda_=lam.cpu().numpy().reshape(B, -1), r=r.cpu().numpy().reshape(B, -1), p=p.cpu().numpy().reshape(B, -1),
               iters=it.cpu().numpy().astype(np.int64), flag=fl.cpu().numpy().astype(np.int64))
    check(out, ob, d, B, ltol=1e-10 if dtype == np.float64 else 1e-6, vtol=1e-9 if dtype == np.float64 else 2e-5)


@pytest.mark.parametrize("n,N,B,dtype", [(18, 128, 5, np.float32), (7, 128, 5, np.float32), (13, 128, 4, np.float64), (24, 128, 1, np.float32),
                                         (3, 128, 6, np.float32), (16, 128, 4, np.float64), (15, 100, 5, np.float32), (5, 300, 4, np.float64),
                                         (2, 600, 5, np.float32), (20, 64, 1, np.float64), (11, 128, 5, np.float32), (9, 40, 7, np.float64)])
def test_round3_shapes_on_the_ill_conditioned_generator(solver, orc, n, N, B, dtype):
    """The kernels instantiated in round 3 (cluster at the odd / small / 16-18 sizes, single-workgroup resident at the odd sizes,
    persistent at 20 and 24) on the a = 0.9 generator: 45-57 iterations instead of 9, where a wrong halo entry or a lost partial shows
    as a different count.  Default path (AUTO), general and default symmetric mode; fp64 counts equal to the oracle's, fp32 within
    the two iterations the summation order moves it -- measured: equal everywhere."""
    d = synth.gen_numpy(n, N, seed=4000 + n + N, batch=B, dtype=dtype, a=0.9)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=300)
    dS, dP, dg = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
    f64 = dtype == np.float64
    for mode in (0, 2):
        solver.set_symmetric(mode)
        try:
            lam = torch.zeros_like(dg)
            it, fl = solver.solve(n, N, B, dS, dP, dg, lam, tol=1e-6, max_iter=300)
            torch.cuda.synchronize()
        finally:
            solver.set_symmetric(2)
        assert not fl.cpu().numpy().any()
        assert np.abs(it.cpu().numpy().astype(np.int64) - ob["iters"].astype(np.int64)).max() <= (0 if f64 else 2)
        lam = lam.cpu().numpy().reshape(B, -1)
        for b in range(B):
            assert relerr(lam[b], ob["lambda_"][b]) < (1e-9 if f64 else 2e-5), (mode, b)


@pytest.mark.parametrize("N,B", [(250, 70), (150, 100), (288, 64), (100, 130)])
def test_cluster_more_problems_than_clusters(solver, orc, N, B):
    """Three- and four-member clusters with more problems than one round holds (64 clusters of four sit 8 blocks apart, 85
    clusters of three next to each other), and a ragged last round of two-member clusters."""
    n = 14
    base = 6
    d = synth.gen_numpy(n, N, seed=900 + N, batch=base, dtype=np.float32)
    idx = np.arange(B) % base
    S, Pi = d["S"][idx], d["Pinv"][idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // base))[:, None]).astype(np.float32)
    ob = orc.pcg_batch(n, N, B, S, Pi, g, tol=1e-6, max_iter=100)
    out = run(solver, n, N, B, S, Pi, g)
    check(out, ob, {"gamma": g}, B)


def test_cluster_declines_what_its_tags_cannot_count(solver, orc):
    """A tag is {launch number, epoch} in 32 bits and a launch may use at most 2^20 epochs: with max_iter = 200000 the
    launch is not eligible and the streaming kernel solves the problems instead -- same answers."""
    n, N, B = 14, 128, 3
    d = synth.gen_numpy(n, N, seed=12, batch=B, dtype=np.float32)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=200000)
    out = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], max_iter=200000)
    check(out, ob, d, B)


def test_cluster_identity_preconditioner(solver, orc):
    """d_Pinv == NULL: r~ = r (pcg.cuh with an identity preconditioner); more iterations, same rules."""
    n, N, B = 14, 128, 4
    d = synth.gen_numpy(n, N, seed=21, batch=B, dtype=np.float32)
    ob = orc.pcg_batch(n, N, B, d["S"], None, d["gamma"], tol=1e-6, max_iter=100)
    out = run(solver, n, N, B, d["S"], None, d["gamma"])
    assert np.abs(out["iters"] - ob["iters"].astype(np.int64)).max() <= 3   # unpreconditioned: order-sensitive (test_gpu_parity.py)
    assert not out["flag"].any()
    for b in range(B):
        assert relerr(out["lambda_"][b], ob["lambda_"][b]) < 2e-5


def test_cluster_takes_what_the_symmetry_test_rejects(solver, orc):
    """Default (tested) symmetric mode: storage that is symmetric goes to the CU-resident symmetric kernel, storage that
    is off by one ulp in one element goes to the cluster kernel -- in one call, each problem equal to the oracle run on the
    very matrices it was given.  (140 problems: a batch one round of clusters holds -- 128 here -- goes to the cluster kernel as a
    whole, without the test; checked with 24 problems as well.)"""
    n, N = 14, 128
    for B in (140, 24):
        d0 = synth.gen_numpy(n, N, seed=611, batch=24, dtype=np.float32)
        idx = np.arange(B) % 24
        d = {"S": d0["S"][idx], "Pinv": d0["Pinv"][idx], "gamma": (d0["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // 24))[:, None]).astype(np.float32)}
        S = unsymmetrize(d["S"], n, N, B, every=3)
        Pinv = unsymmetrize(d["Pinv"], n, N, B, every=4)
        ob = orc.pcg_batch(n, N, B, S, Pinv, d["gamma"], tol=1e-6, max_iter=100, nthreads=8)
        out = run(solver, n, N, B, S, Pinv, d["gamma"], symmetric=2)
        check(out, ob, d, B)


@pytest.mark.parametrize("tol,max_iter", [(1e-6, 0), (1e-6, 1), (1e30, 5), (0.0, 2), (0.0, 7)])
def test_cluster_iteration_edges(solver, orc, tol, max_iter):
    n, N, B = 14, 128, 3
    d = synth.gen_numpy(n, N, seed=78, batch=B, dtype=np.float32)
    lam0 = (np.stack([synth.normals(5 + b, 0, n * N) for b in range(B)]) * 0.1).astype(np.float32)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], lambda0=lam0, tol=tol, max_iter=max_iter)
    out = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], lam0=lam0, tol=tol, max_iter=max_iter)
    assert np.array_equal(out["iters"], ob["iters"].astype(np.int64))
    assert np.array_equal(out["flag"], ob["max_iter_exit"].astype(np.int64))
    scale = np.abs(d["gamma"]).max()
    for key in ("lambda_", "r", "p"):
        assert np.abs(out[key] - ob[key].reshape(B, -1)).max() < 2e-5 * max(scale, np.abs(ob[key]).max()), key


def test_cluster_full_config3_batch(solver, orc):
    """BASELINE config 3's batch (1024 problems, n = 14, N = 128) in general storage: eight rounds of 128 clusters.
    Iteration counts in the band of the generator, true residuals small, and 64 problems spread over the rounds equal to
    the oracle."""
    n, N, B = 14, 128, 1024
    d = synth.gen_numpy(n, N, seed=1234, batch=B, dtype=np.float32)
    out = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], max_iter=25)
    assert out["iters"].min() >= 7 and out["iters"].max() <= 12 and not out["flag"].any()
    idx = np.arange(0, B, 16)
    ob = orc.pcg_batch(n, N, len(idx), d["S"][idx], d["Pinv"][idx], d["gamma"][idx], tol=1e-6, max_iter=25)
    assert np.array_equal(out["iters"][idx], ob["iters"].astype(np.int64))
    for j, b in enumerate(idx):
        assert relerr(out["lambda_"][b], ob["lambda_"][j]) < 1e-6
    for b in range(0, B, 37):   # true residual, fp64 product of the oracle
        res = d["gamma"][b].astype(np.float64) - orc.spmv(n, N, d["S"][b].astype(np.float64), out["lambda_"][b].astype(np.float64))
        assert np.linalg.norm(res) < 2e-3 * np.linalg.norm(d["gamma"][b])


def test_cluster_config5_batch_on_one_gpu(solver, orc):
    """BASELINE config 5's 8192 problems (generated on the device with the seeds 1234 + i of SURVEY.md section 8d) in general
    storage on one GPU: 64 rounds of 128 clusters in one launch.  Every problem converges in 7..12 iterations, the true
    residual is small, a warm restart exits after one iteration, and 64 problems from both ends and the middle of the
    batch equal the oracle run on the same device-formed Pinv."""
    n, N, B = 14, 128, 8192
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32, seed=1234)
    S, gamma = g["S"], g["gamma"]
    del g
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    r, p = torch.empty_like(gamma), torch.empty_like(gamma)
    solver.set_symmetric(0)
    try:
        iters, flags = solver.solve(n, N, B, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)
        torch.cuda.synchronize()
        it = iters.cpu().numpy()
        assert flags.sum().item() == 0 and it.min() >= 7 and it.max() <= 12, (it.min(), it.max())
        res = gamma - solver.spmv(n, N, B, S, lam)
        assert (res.norm(dim=1) / gamma.norm(dim=1)).max().item() < 2e-3
        assert ((r - res).norm(dim=1) / gamma.norm(dim=1)).max().item() < 1e-5
        idx = np.r_[0:24, 4090:4106, 8168:8192]
        ob = orc.pcg_batch(n, N, len(idx), S[idx].cpu().numpy(), P[idx].cpu().numpy(), gamma[idx].cpu().numpy(),
                           tol=1e-6, max_iter=25, nthreads=8)
        assert np.array_equal(it[idx], ob["iters"].astype(np.int64))
        lam_h = lam[idx].cpu().numpy()
        for k in range(len(idx)):
            assert relerr(lam_h[k], ob["lambda_"][k]) < 1e-6, (idx[k], it[idx[k]])
        iters2, flags2 = solver.solve(n, N, B, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)   # warm restart
        torch.cuda.synchronize()
        assert int(iters2.max()) == 1 and int(iters2.min()) == 1 and flags2.sum().item() == 0
    finally:
        solver.set_symmetric(2)


def test_cluster_replays_need_no_clearing(solver, orc):
    """Every launch starts its epochs at 1 and nothing is cleared between launches (a tag carries the launch's number, so
    what an earlier launch left in a slot is never taken for a publication): a graph replayed back to back, other shapes
    and iteration limits in between -- always the same answer."""
    n, N, B = 14, 128, 200
    d = synth.gen_numpy(n, N, seed=99, batch=B, dtype=np.float32)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=25)
    first = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], max_iter=25)
    assert np.array_equal(first["iters"], ob["iters"].astype(np.int64))
    d2 = synth.gen_numpy(n, 200, seed=98, batch=3, dtype=np.float32)
    run(solver, n, 200, 3, d2["S"], d2["Pinv"], d2["gamma"], tol=0.0, max_iter=3)   # three members, fixed iterations
    solver.set_symmetric(0)
    try:
        dS, dP, dg = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
        lam = torch.zeros_like(dg)
        r, p = torch.empty_like(dg), torch.empty_like(dg)
        it = torch.zeros(B, dtype=torch.int32, device="cuda")
        fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
        gr = solver.graph_solve(n, N, B, dS, dP, dg, lam, r, p, 1e-6, 25, it, fl)
        for _ in range(30):
            lam.zero_()
            gr.launch()
        torch.cuda.synchronize()
        gr.close()
    finally:
        solver.set_symmetric(2)
    assert np.array_equal(it.cpu().numpy().astype(np.int64), first["iters"])
    assert np.array_equal(lam.cpu().numpy().reshape(B, -1), first["lambda_"])   # same kernel, same inputs: same bits
    again = run(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], max_iter=25)
    assert np.array_equal(again["lambda_"], first["lambda_"])


def test_cluster_launches_do_not_see_each_other(solver, orc):
    """Two launches that use the same epochs for different problems: the first solves three problems per cluster, the second
    only the LAST quarter of a larger batch (the rest is symmetric and goes to the other kernel), after a kernel sequence
    that may put the workgroups on other XCDs.  (With tags that were bare epochs and slots cleaned by their owners, this
    sequence once produced answers built from stale granules on one box; a tag now carries the launch's number.)"""
    n, N, mi = 14, 128, 3
    B1, B2 = 384, 512        # 3 and 4 rounds of 128 clusters
    d = synth.gen_numpy(n, N, seed=31, batch=8, dtype=np.float32)
    idx1, idx2 = np.arange(B1) % 8, np.arange(B2) % 8
    run(solver, n, N, B1, d["S"][idx1], d["Pinv"][idx1], d["gamma"][idx1], tol=0.0, max_iter=mi)          # all general
    S2, P2, g2 = d["S"][idx2].copy(), d["Pinv"][idx2].copy(), d["gamma"][idx2]
    S2[384:] = unsymmetrize(S2[384:], n, N, 128)                                                        # only these are general
    out = run(solver, n, N, B2, S2, P2, g2, tol=0.0, max_iter=mi, symmetric=2)
    sub = np.arange(384, 512, 9)
    ob = orc.pcg_batch(n, N, len(sub), S2[sub], P2[sub], g2[sub], tol=0.0, max_iter=mi)
    assert (out["iters"] == mi).all() and (out["flag"] == 1).all()
    for j, b in enumerate(sub):
        assert relerr(out["lambda_"][b], ob["lambda_"][j]) < 1e-6, b


HOOKS_LIB = os.path.join(ROOT, "gbd-pcg_amd", "csrc", "variants", "libgbdpcg_hooks.so")

GIVE_UP = r"""
import os, numpy as np, torch
from gbd_pcg_amd import binding, synth
from oracle import oracle as orc
n, N, B = int(os.environ.get("GU_N", "14")), int(os.environ.get("GU_KNOTS", "128")), 300
dt = np.float64 if os.environ.get("GU_DT", "f32") == "f64" else np.float32
d = synth.gen_numpy(n, N, seed=5, batch=B, dtype=dt)
s = binding.Solver(0)
s.set_symmetric(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dS, dP, dg = t(d["S"]), t(d["Pinv"]), t(d["gamma"])
H = s.cluster_members(np.dtype(dt).itemsize, n, N)
assert H >= 1
clusters = 256 // H
# block 1: member 0 of cluster 1 where the members sit 8 blocks apart (cluster count a multiple of 8), else a member of cluster 1 // H
lost = np.arange(B) % clusters == (1 if clusters % 8 == 0 else 1 // H)
rescued = "GBDPCG_RESCUE_OFF" not in os.environ
for rnd in range(2):   # the second launch (same hook: the dropped workgroup is a property of the process) is as good as the first
    lam = torch.full_like(dg, 0.25)
    lam0 = lam.clone()
    r, p = torch.full_like(dg, 7.0), torch.full_like(dg, 7.0)
    it, fl = s.solve(n, N, B, dS, dP, dg, lam, r, p, tol=1e-6, max_iter=25)
    torch.cuda.synchronize()
    it = it.cpu().numpy().astype(np.int64) & 0xffffffff; fl = fl.cpu().numpy()
    if not rescued:
        # what the cluster kernel leaves behind when a cluster cannot meet: the mark, and the caller's buffers untouched
        assert (it[lost] == 0xffffffff).all() and (fl[lost] == 2).all(), (it[lost], fl[lost])
        lb = torch.from_numpy(lost).cuda()
        assert torch.equal(lam[lb], lam0[lb]) and bool((r[lb] == 7.0).all()) and bool((p[lb] == 7.0).all())
        assert (it[~lost] < 20).all() and (fl[~lost] == 0).all()
print("GIVE-UP-SEEN" if not rescued else "")
if rescued:
    # with the rescue launch nobody sees a mark: every problem, the dropped cluster's included, equals the oracle's solve
    # from the same initial guess
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], lambda0=np.full((B, n * N), 0.25, dt), tol=1e-6,
                       max_iter=25, nthreads=8)
    assert np.array_equal(it, ob["iters"].astype(np.int64)), (it[lost], ob["iters"][lost])
    assert np.array_equal(fl.astype(bool), ob["max_iter_exit"])
    lam = lam.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(lam - ob["lambda_"], axis=1) / np.linalg.norm(ob["lambda_"], axis=1)
    assert err.max() < (1e-10 if dt == np.float64 else 1e-6), (err.max(), err[lost].max())
    scale = np.abs(d["gamma"]).max(axis=1)
    vt = 1e-9 if dt == np.float64 else 2e-5
    assert (np.abs(r.cpu().numpy() - ob["r"]).max(axis=1) < vt * scale).all()
    assert (np.abs(p.cpu().numpy() - ob["p"]).max(axis=1) < vt * scale).all()
    print("RESCUED-OK", int(lost.sum()))
"""


def _run_hooked(script, **env_extra):
    assert os.path.exists(HOOKS_LIB), "make -C gbd-pcg_amd/csrc builds variants/libgbdpcg_hooks.so"
    env = dict(os.environ, GBDPCG_LIB=HOOKS_LIB, PYTHONPATH=ROOT, **env_extra)
    return subprocess.run([sys.executable, "-c", script], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_cluster_gives_up_instead_of_hanging():
    """A workgroup that never publishes (fault injection of variants/libgbdpcg_hooks.so; the shipped library has no such
    hook): its cluster ends after a bounded spin and marks its problems (max_iter_exit = 2, iters = 0xffffffff, lambda /
    r / p untouched), every other cluster is unaffected, and the next launch is not disturbed by what the broken one
    left behind.  Seen with the rescue launch switched off (GBDPCG_RESCUE_OFF, hooks build only)."""
    out = _run_hooked(GIVE_UP, GBDPCG_CLUSTER_DROP_WG="1", GBDPCG_CLUSTER_SPIN_LIMIT="20000", GBDPCG_RESCUE_OFF="1")
    assert out.returncode == 0 and "GIVE-UP-SEEN" in out.stdout, out.stdout + out.stderr


def test_cluster_give_up_is_rescued():
    """The same fault with the library's default behaviour: the streaming launch queued behind the cluster kernel solves
    the problems that carry the mark, so AUTO never hands back an invalid result -- iteration counts, flags, lambda, r
    and p of EVERY problem equal the oracle's (what the reference guarantees by refusing a launch that cannot be
    co-resident before it starts, /root/reference/include/pcg.cuh:23-49)."""
    out = _run_hooked(GIVE_UP, GBDPCG_CLUSTER_DROP_WG="1", GBDPCG_CLUSTER_SPIN_LIMIT="20000")
    assert out.returncode == 0 and "RESCUED-OK 3" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n,N,dt,lost", [(18, 128, "f32", 4), (13, 128, "f64", 5), (16, 64, "f32", 2), (7, 200, "f32", 4), (16, 128, "f64", 5)])
def test_cluster_give_up_is_rescued_at_the_other_shapes(n, N, dt, lost):
    """The same at shapes of round 3: three members with the D and R blocks of Pinv in LDS (18), four members in fp64 (13, 16), a
    cluster of one (16 x 64: the silenced workgroup IS its cluster) and a small odd size; `lost` problems of the 300 sit in the
    silenced cluster and come back solved by the workgroup that leaves it last."""
    out = _run_hooked(GIVE_UP, GBDPCG_CLUSTER_DROP_WG="1", GBDPCG_CLUSTER_SPIN_LIMIT="20000", GU_N=str(n), GU_KNOTS=str(N), GU_DT=dt)
    assert out.returncode == 0 and "RESCUED-OK %d" % lost in out.stdout, out.stdout + out.stderr
