"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (BASELINE.json north_star "lambda within 1e-6 relative of the
reference at equal iteration count"):
  fp64 : ||lambda - lambda_oracle||_2 / ||lambda_oracle||_2 <= 1e-10, equal iteration count
  fp32 : <= 1e-6 norm-wise vs the fp32 oracle at equal iteration count (well-conditioned
         generator, a = 0.5), and error vs the fp64 oracle <= 2x the fp32 oracle's own error
  SpMV : fp64 <= 1e-13, fp32 <= 1e-6 norm-wise vs the dense fp64 product
Bit-exactness is not claimed: the summation order inside the reference's GLASS primitives is
unknown ("parity unpinned" at the last bit, oracle/pcg_oracle.c).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from gbd_pcg_amd import binding, synth  # noqa: E402

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10
F32_TOL = 1e-6
PATHS = [binding.PATH_FUSED, binding.PATH_SPLIT]
PATH_NAME = {binding.PATH_FUSED: "fused", binding.PATH_SPLIT: "split"}


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
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_solve(solver, n, N, batch, S, Pinv, gamma, lam0=None, tol=1e-6, max_iter=100, path=binding.PATH_AUTO):
    solver.set_path(path)
    dS, dg = dev(S), dev(gamma)
    dP = None if Pinv is None else dev(Pinv)
    lam = torch.zeros_like(dg) if lam0 is None else dev(lam0)
    r, p = torch.empty_like(dg), torch.empty_like(dg)
    iters, flags = solver.solve(n, N, batch, dS, dP, dg, lam, r, p, tol=tol, max_iter=max_iter)
    torch.cuda.synchronize()
    solver.set_path(binding.PATH_AUTO)
    return dict(lambda_=lam.cpu().numpy().reshape(batch, -1), r=r.cpu().numpy().reshape(batch, -1),
                p=p.cpu().numpy().reshape(batch, -1), iters=iters.cpu().numpy().astype(np.int64),
                max_iter_exit=flags.cpu().numpy().astype(bool))


# ------------------------------------------------------------------------------------- SpMV
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,N,batch", [(2, 3, 1), (3, 5, 2), (7, 2, 3), (14, 1, 2), (5, 9, 4), (14, 64, 1),
                                       (14, 128, 8), (36, 256, 1), (16, 33, 2), (12, 40, 3), (64, 5, 1),
                                       (37, 6, 2)])
def test_spmv_vs_oracle_and_dense(solver, orc, dtype, n, N, batch):
    d = synth.gen_numpy(n, N, seed=50, batch=batch, dtype=dtype)
    x = np.stack([synth.normals(60 + b, 0, n * N) for b in range(batch)]).astype(dtype)
    y = solver.spmv(n, N, batch, dev(d["S"]), dev(x))
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    yo = orc.spmv(n, N, d["S"], x, batch=batch).reshape(batch, -1)
    tol = 1e-13 if dtype == np.float64 else F32_TOL
    for b in range(batch):
        A = orc.dense_from_bt(n, N, d["S"][b])
        assert relerr(y[b], A @ x[b].astype(np.float64)) < tol
        assert relerr(y[b], yo[b]) < tol


def test_spmv_ignores_unused_corner_blocks(solver):
    """L_0 / R_{N-1} are never read (pcg.cuh:105-106): NaNs there must not reach y."""
    n, N, batch = 14, 6, 3
    d = synth.gen_numpy(n, N, seed=51, batch=batch, dtype=np.float32)
    x = np.stack([synth.normals(70 + b, 0, n * N) for b in range(batch)]).astype(np.float32)
    y0 = solver.spmv(n, N, batch, dev(d["S"]), dev(x)).cpu().numpy()
    S = d["S"].copy()
    S[:, : n * n] = np.nan
    S[:, -n * n:] = np.nan
    y1 = solver.spmv(n, N, batch, dev(S), dev(x)).cpu().numpy()
    assert np.array_equal(y0, y1)


def test_spmv_linearity_full_size(solver):
    """BASELINE config 3 at full size (n=14, N=128, batch=1024): S(ax + by) = a Sx + b Sy."""
    n, N, batch = 14, 128, 1024
    g = synth.gen_torch(n, N, batch, "cuda", torch.float32, seed=5)
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((batch, n * N), device="cuda", generator=gen)
    z = torch.randn((batch, n * N), device="cuda", generator=gen)
    lhs = solver.spmv(n, N, batch, g["S"], (0.5 * x - 2.0 * z).contiguous())
    rhs = 0.5 * solver.spmv(n, N, batch, g["S"], x) - 2.0 * solver.spmv(n, N, batch, g["S"], z)
    torch.cuda.synchronize()
    err = (lhs - rhs).norm(dim=1) / rhs.norm(dim=1)
    assert float(err.max()) < 2e-6
    # symmetry of S: x.(S z) == z.(S x)
    a = (x.double() * solver.spmv(n, N, batch, g["S"], z).double()).sum(1)
    b = (z.double() * solver.spmv(n, N, batch, g["S"], x).double()).sum(1)
    assert float(((a - b).abs() / (a.abs() + b.abs() + 1e-30)).max()) < 1e-4


# ------------------------------------------------------------------------------------- PCG
@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
def test_readme_system_fp64(solver, orc, golden_dir, path):
    """BASELINE config 1 shape (n=2, N=3, fp64): the reference's example system
    (examples/pcg_solve_dp.cu:14-25) -> dense answer, 6 iterations with Pinv = I, 3 with the stair."""
    G = np.load(os.path.join(golden_dir, "readme.npz"))
    n, N, S, gamma = orc.readme_system()
    out = gpu_solve(solver, n, N, 1, S, None, gamma, tol=1e-6, max_iter=25, path=path)
    assert out["iters"][0] == 6 == int(G["iters_f64_ident"]) and not out["max_iter_exit"][0]
    assert relerr(out["lambda_"][0], G["lambda_dense"]) < 1e-10
    assert relerr(out["lambda_"][0], G["lambda_f64_ident"]) < F64_TOL
    L, D, R = synth.unpack_bt(n, N, S)
    P = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    out = gpu_solve(solver, n, N, 1, S, P, gamma, tol=1e-6, max_iter=25, path=path)
    assert out["iters"][0] == 3 and relerr(out["lambda_"][0], G["lambda_f64_stair"]) < F64_TOL


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
def test_readme_system_fp32(solver, orc, path):
    """fp32 twin (examples/pcg_solve.cu:14-25) with the stair preconditioner: 3 iterations under
    every summation order; kappa ~ 1562 so the answer is good to ~1e-5."""
    n, N, S, gamma = orc.readme_system()
    L, D, R = synth.unpack_bt(n, N, S)
    P = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    lam = np.linalg.solve(orc.dense_from_bt(n, N, S), gamma)
    out = gpu_solve(solver, n, N, 1, S.astype(np.float32), P.astype(np.float32), gamma.astype(np.float32),
                    max_iter=25, path=path)
    assert out["iters"][0] == 3 and relerr(out["lambda_"][0], lam) < 5e-5


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
@pytest.mark.parametrize("name", ["gen_2x3", "gen_3x5", "gen_7x2", "gen_14x1", "gen_14x64", "gen_14x128",
                                  "gen_36x256"])
@pytest.mark.parametrize("pinv", ["stair", "ident"])
def test_golden_fp64(solver, golden_dir, path, name, pinv):
    """Committed fixtures (tests/golden): fp64 lambda <= 1e-10 of the oracle's, equal iterations.
    gen_14x64 / gen_14x128 / gen_36x256 are BASELINE configs 2 / 3 (one problem) / 4."""
    G = np.load(os.path.join(golden_dir, name + ".npz"))
    n, N = int(G["n"]), int(G["N"])
    d = synth.gen_numpy(n, N, seed=int(G["seed"]), a=float(G["a"]))
    P = d["Pinv"] if pinv == "stair" else None
    out = gpu_solve(solver, n, N, 1, d["S"], P, d["gamma"], tol=1e-6, max_iter=100, path=path)
    assert out["iters"][0] == int(G[f"iters_f64_{pinv}"]) and not out["max_iter_exit"][0]
    assert relerr(out["lambda_"][0], G[f"lambda_f64_{pinv}"]) < F64_TOL


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
@pytest.mark.parametrize("n,N", [(14, 64), (14, 128), (36, 256), (3, 5), (2, 3)])
def test_fp32_vs_oracle(solver, orc, golden_dir, path, n, N):
    """fp32: <= 1e-6 norm-wise vs the fp32 oracle at equal iteration count, and no worse than 2x the
    fp32 oracle's own distance from the fp64 oracle (BASELINE configs 2, 3 (one problem), 4 in fp32)."""
    G = np.load(os.path.join(golden_dir, f"gen_{n}x{N}.npz"))
    d = synth.gen_numpy(n, N, seed=1234, a=0.5, dtype=np.float32)
    o32 = orc.pcg(n, N, d["S"][0], d["Pinv"][0], d["gamma"][0], tol=1e-6, max_iter=100)
    out = gpu_solve(solver, n, N, 1, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100, path=path)
    assert out["iters"][0] == o32["iters"] == int(G["iters_f32_stair"])
    assert relerr(out["lambda_"][0], o32["lambda_"]) < F32_TOL
    ref64 = G["lambda_f64_stair"]
    assert relerr(out["lambda_"][0], ref64) < 2 * relerr(o32["lambda_"], ref64) + 1e-7
    # final r and p are left behind for the caller (pcg.cuh:175,205); at convergence they are
    # rounding-sized, so compare on the scale of gamma
    gnorm = np.linalg.norm(d["gamma"][0])
    assert np.linalg.norm(out["r"][0] - o32["r"]) < 1e-5 * gnorm
    assert np.linalg.norm(out["p"][0] - o32["p"]) < 1e-5 * gnorm


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 16, 18, 20, 24, 36, 37, 48])
def test_state_sizes(solver, orc, path, dtype, n):
    """Every block size: the compile-time specialised kernels (2, 4, 6, 8, 10, 12, 13, 14, 16, 18, 20, 24, 36)
    and the runtime-n pipeline (the rest) against the oracle, lambda and iteration count."""
    N, B = 11, 3
    d = synth.gen_numpy(n, N, seed=300 + n, batch=B, dtype=dtype)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=60)
    out = gpu_solve(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=60, path=path)
    assert np.array_equal(out["iters"], ob["iters"]) and not out["max_iter_exit"].any()
    tol = F64_TOL if dtype == np.float64 else F32_TOL
    for b in range(B):
        assert relerr(out["lambda_"][b], ob["lambda_"][b]) < tol


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_batch_of_config3(solver, orc, golden_dir, path, dtype):
    """G6: eight problems of BASELINE config 3's batch (seeds 1234+i), solved as one batch."""
    G = np.load(os.path.join(golden_dir, "gen_14x128_batch8.npz"))
    n, N, B = 14, 128, 8
    d = synth.gen_numpy(n, N, seed=1234, batch=B, dtype=dtype)
    out = gpu_solve(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100, path=path)
    assert np.array_equal(out["iters"], G["iters_f64_stair"]) and not out["max_iter_exit"].any()
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100)
    for b in range(B):
        if dtype == np.float64:
            assert relerr(out["lambda_"][b], G["lambda_f64_stair"][b]) < F64_TOL
        else:
            assert relerr(out["lambda_"][b], ob["lambda_"][b]) < F32_TOL


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fixed_iteration_state(solver, orc, path, dtype):
    """exit_tol = 0, max_iter = k: exactly k iterations (abs(eta) < 0 never holds, pcg.cuh:195),
    max_iter_exit set, and lambda, r, p all match the oracle's state after k iterations."""
    n, N, B = 14, 20, 3
    d = synth.gen_numpy(n, N, seed=77, batch=B, dtype=dtype)
    for k in (0, 1, 2, 5):
        out = gpu_solve(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=k, path=path)
        ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=k)
        assert (out["iters"] == k).all() and out["max_iter_exit"].all()
        tol = 1e-10 if dtype == np.float64 else 1e-5
        for key in ("lambda_", "r", "p"):
            for b in range(B):
                # r and p shrink as the solve proceeds: measure them on the scale of gamma
                scale = max(np.linalg.norm(ob[key][b]), 1e-2 * np.linalg.norm(d["gamma"][b]))
                if key == "lambda_" and k == 0:
                    assert not out[key][b].any()
                    continue
                assert np.linalg.norm(out[key][b].astype(np.float64) - ob[key][b]) < tol * scale, (k, key, b)


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
def test_ragged_convergence_in_one_batch(solver, orc, path):
    """Problems of one batch exit at their own iteration counts: a warm-started problem (1
    iteration), a harder one (a = 0.9) and a capped one share a launch."""
    n, N = 14, 24
    easy = synth.gen_numpy(n, N, seed=5, a=0.5)
    hard = synth.gen_numpy(n, N, seed=6, a=0.9)
    S = np.concatenate([easy["S"], hard["S"], easy["S"]])
    P = np.concatenate([easy["Pinv"], hard["Pinv"], easy["Pinv"]])
    g = np.concatenate([easy["gamma"], hard["gamma"], easy["gamma"]])
    lam_star = np.linalg.solve(orc.dense_from_bt(n, N, easy["S"][0]), easy["gamma"][0])
    lam0 = np.zeros_like(g)
    lam0[2] = lam_star
    out = gpu_solve(solver, n, N, 3, S, P, g, lam0=lam0, tol=1e-6, max_iter=20, path=path)
    ob = [orc.pcg(n, N, S[b], P[b], g[b], lambda0=lam0[b], tol=1e-6, max_iter=20) for b in range(3)]
    assert [o["iters"] for o in ob] == list(out["iters"])
    assert out["iters"][2] == 1 and out["iters"][1] == 20 and out["max_iter_exit"][1]
    assert not out["max_iter_exit"][0] and not out["max_iter_exit"][2]
    for b in range(3):
        assert relerr(out["lambda_"][b], ob[b]["lambda_"]) < F64_TOL


# DenseGeom::MAX_KNOTS = 8 waves x floor(64 / lanes per knot); fp32 with an even block size runs two rows per lane, everything else one
RES_MAX = {np.float32: {2: 512, 4: 256, 6: 168, 8: 128, 10: 96, 12: 80, 3: 168, 5: 96, 7: 72, 9: 56, 11: 40, 13: 32, 15: 32},
           np.float64: {2: 256, 4: 128, 6: 80, 8: 64, 10: 48, 12: 40, 3: 168, 5: 96, 7: 72, 9: 56, 11: 40, 13: 32}}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [2, 4, 6, 8, 10, 12, 3, 5, 7, 9, 11, 13, 15])
def test_register_resident_kernel_of_the_small_blocks(solver, orc, dtype, n):
    """pcg_resident_kernel is built for n in {2 (from 16 knots on), 3, ..., 13} as well as 14 (fp32 at even sizes: two rows per lane, else one): both matrices
    in the registers of one workgroup for the whole solve, several workgroups per compute unit.  Against the oracle at the
    longest horizon the block size allows, one knot beyond it (the streaming kernel takes over: same answers), one knot,
    batches larger than the grid (every workgroup walks several problems), with and without a preconditioner, from a warm
    start, and with r / p checked after a fixed number of iterations."""
    if n == 15 and dtype == np.float64:
        pytest.skip("fp64 at stateSize 15 is the cluster kernel's (180 matrix registers)")
    tol = F64_TOL if dtype == np.float64 else F32_TOL
    top = RES_MAX[dtype][n]
    for N, B, pinv in ((top, 3, True), (top + 1, 2, True), (1, 5, True), (top // 2 + 1, 3, False), (9, 1300, True)):
        d = synth.gen_numpy(n, N, seed=900 + n + N, batch=B, dtype=dtype)
        P = d["Pinv"] if pinv else None
        ob = orc.pcg_batch(n, N, B, d["S"], P, d["gamma"], tol=1e-6, max_iter=200)
        out = gpu_solve(solver, n, N, B, d["S"], P, d["gamma"], tol=1e-6, max_iter=200)
        if pinv:
            assert np.array_equal(out["iters"], ob["iters"]) and not out["max_iter_exit"].any(), (N, B)
        else:   # no preconditioner: the count is sensitive to the summation order (see test_randomized_dispatch_sweep)
            assert (np.abs(out["iters"] - ob["iters"]) <= 3).all() and not out["max_iter_exit"].any(), (N, B)
        for b in range(0, B, max(1, B // 7)):
            # (no preconditioner: hundreds of iterations on the long horizons of the small blocks, each with its own summation
            # order in the two implementations -- the solutions agree to the conditioning of S, not to the last digits)
            assert relerr(out["lambda_"][b], ob["lambda_"][b]) < (tol if pinv else 100 * tol), (N, B, b)
    # fixed iteration count: lambda, r and p after exactly 4 iterations, from a warm start
    N, B = top - 3, 4
    d = synth.gen_numpy(n, N, seed=77 + n, batch=B, dtype=dtype)
    lam0 = (0.1 * np.random.default_rng(n).standard_normal((B, n * N))).astype(dtype)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=4, lambda0=lam0)
    out = gpu_solve(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], lam0=lam0, tol=0.0, max_iter=4)
    assert (out["iters"] == 4).all() and out["max_iter_exit"].all()
    for k in ("lambda_", "r", "p"):
        for b in range(B):
            assert relerr(out[k][b], ob[k][b]) < 20 * tol, (k, b)
    # ... and the same without a preconditioner: the loose band above (exit iteration within 3, lambda to the
    # conditioning of S) would let a halo or summation bug of the Pinv = NULL branch through; four fixed iterations do not
    # depend on where the exit test fires, so lambda, r and p are held to the same 20 * tol as with Pinv
    ob = orc.pcg_batch(n, N, B, d["S"], None, d["gamma"], tol=0.0, max_iter=4, lambda0=lam0)
    out = gpu_solve(solver, n, N, B, d["S"], None, d["gamma"], lam0=lam0, tol=0.0, max_iter=4)
    assert (out["iters"] == 4).all() and out["max_iter_exit"].all()
    for k in ("lambda_", "r", "p"):
        for b in range(B):
            assert relerr(out[k][b], ob[k][b]) < 20 * tol, (k, b)
    assert solver.choose_path(np.dtype(dtype).itemsize, n, top, 1) == binding.PATH_FUSED
    # the first launch of a shape may be the capture of a graph (the workgroups-per-CU query happens inside it)
    N, B = max(2, top // 3), 7
    d = synth.gen_numpy(n, N, seed=5 + n, batch=B, dtype=dtype)
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=100)
    dS, dP, dg = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
    lam = torch.zeros_like(dg)
    r, p = torch.empty_like(dg), torch.empty_like(dg)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    gr = solver.graph_solve(n, N, B, dS, dP, dg, lam, r, p, 1e-6, 100, it, fl)
    for _ in range(2):
        lam.zero_()
        gr.launch()
        torch.cuda.synchronize()
        assert np.array_equal(it.cpu().numpy().astype(np.int64), ob["iters"].astype(np.int64))
        assert relerr(lam.cpu().numpy().reshape(B, -1)[B - 1], ob["lambda_"][B - 1]) < tol
    gr.close()


def test_auto_path_choice(solver):
    """BASELINE configs: 2 and 3 run fused (vectors fit one workgroup's LDS); 4 is spread over many CUs -- as ONE
    persistent launch when all its workgroups can be resident (one problem, or two), else two launches per iteration."""
    assert solver.choose_path(4, 14, 64, 1) == binding.PATH_FUSED
    assert solver.choose_path(4, 14, 128, 1024) == binding.PATH_FUSED
    assert solver.choose_path(8, 36, 256, 1) == binding.PATH_PERSISTENT
    assert solver.choose_path(8, 36, 256, 2) == binding.PATH_PERSISTENT
    assert solver.choose_path(8, 36, 256, 16) == binding.PATH_SPLIT
    assert solver.choose_path(8, 36, 64, 1) == binding.PATH_PERSISTENT    # fits one workgroup, but would stream 2 MB per iteration through one CU


def test_blocking_and_host_overloads(solver, orc, golden_dir):
    """The two reference entry points: device-pointer overload (interface.cuh:92-144) returns the
    iteration count; host overload (interface.cuh:24-89) with Pinv = NULL means identity."""
    G = np.load(os.path.join(golden_dir, "readme.npz"))
    n, N, S, gamma = orc.readme_system()
    dS, dg = dev(S), dev(gamma)
    lam = torch.zeros_like(dg)
    it, flag = solver.solve_blocking(n, N, dS, None, dg, lam, tol=1e-6, max_iter=25)
    assert it == 6 and not flag and relerr(lam.cpu().numpy(), G["lambda_dense"]) < 1e-10
    h_lam = np.zeros(n * N)
    it, flag = solver.solve_host(n, N, S, None, gamma, h_lam, tol=1e-6, max_iter=25)
    assert it == 6 and not flag and relerr(h_lam, G["lambda_dense"]) < 1e-10


@pytest.mark.parametrize("path", PATHS, ids=PATH_NAME.get)
def test_graph_replay(solver, orc, path):
    """hipGraph-captured solve: replay gives the same lambda as the eager launch, every time."""
    n, N, B = 14, 16, 4
    d = synth.gen_numpy(n, N, seed=31, batch=B, dtype=np.float64)
    eager = gpu_solve(solver, n, N, B, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=30, path=path)
    solver.set_path(path)
    dS, dP, dg = dev(d["S"]), dev(d["Pinv"]), dev(d["gamma"])
    lam = torch.zeros_like(dg)
    r, p = torch.empty_like(dg), torch.empty_like(dg)
    iters = torch.zeros(B, dtype=torch.int32, device="cuda")
    flags = torch.zeros(B, dtype=torch.uint8, device="cuda")
    g = solver.graph_solve(n, N, B, dS, dP, dg, lam, r, p, 1e-6, 30, iters, flags)
    solver.set_path(binding.PATH_AUTO)
    for _ in range(3):
        lam.zero_()
        g.launch()
        torch.cuda.synchronize()
        assert np.array_equal(lam.cpu().numpy().reshape(B, -1), eager["lambda_"])
        assert np.array_equal(iters.cpu().numpy(), eager["iters"])
    g.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["stair", "jacobi", "identity"])
@pytest.mark.parametrize("n", [14, 36])
def test_form_pinv(solver, dtype, kind, n):
    """f1: Pinv built on the device from S equals the host construction (n = 36: one column per lane, in place)."""
    N, B = 9, 3
    d = synth.gen_numpy(n, N, seed=41, batch=B, dtype=dtype, pinv=kind)
    code = {"stair": binding.PINV_STAIR, "jacobi": binding.PINV_BLOCK_JACOBI, "identity": binding.PINV_IDENTITY}[kind]
    P = solver.form_pinv(n, N, B, dev(d["S"]), code)
    torch.cuda.synchronize()
    P = P.cpu().numpy().reshape(B, N, 3, n * n).copy()
    want = d["Pinv"].reshape(B, N, 3, n * n).copy()
    # the never-read corner slots are unspecified
    for arr in (P, want):
        arr[:, 0, 0] = 0
        arr[:, -1, 2] = 0
    assert relerr(P, want) < (1e-12 if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,N,B", [(14, 2, 2), (14, 15, 1), (14, 16, 3), (14, 17, 2), (14, 31, 1), (14, 128, 2),
                                   (8, 20, 2), (12, 33, 1), (16, 47, 2), (6, 5, 2), (13, 18, 1), (36, 4, 1), (36, 1, 2),
                                   (36, 2, 1), (36, 21, 3), (3, 19, 3), (3, 2, 1), (5, 33, 2), (7, 18, 2), (9, 21, 1), (11, 17, 2), (15, 16, 2),
                                   (15, 1, 1), (20, 9, 2), (22, 13, 2), (22, 2, 1), (24, 7, 1), (18, 10, 2), (10, 40, 1), (4, 64, 2), (2, 30, 3)])
def test_form_pinv_shapes(solver, dtype, n, N, B):
    """The stair across the kernel families (fused 16-knot workgroups with their chunk seams at 15 / 16 / 17 knots,
    register Gauss-Jordan, LDS forms) against the host construction; exact symmetry wherever S is symmetric."""
    d = synth.gen_numpy(n, N, seed=300 + n + N, batch=B, dtype=dtype)
    P = solver.form_pinv(n, N, B, dev(d["S"]), binding.PINV_STAIR)
    torch.cuda.synchronize()
    assert solver.check_symmetric(n, N, B, P).cpu().numpy().tolist() == [1] * B
    P = P.cpu().numpy().reshape(B, N, 3, n * n).copy()
    want = d["Pinv"].reshape(B, N, 3, n * n).copy()
    for arr in (P, want):
        arr[:, 0, 0] = 0
        arr[:, -1, 2] = 0
    assert relerr(P, want) < (1e-12 if dtype == np.float64 else 2e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [14, 36])
def test_form_pinv_general_S(solver, dtype, n):
    """The stair of a NON-symmetric S: the device evaluates the left slot of a knot from L_{k+1} itself whenever
    it is not the mirror image of R_k (the mirror shortcut is for symmetric storage only).  One problem keeps
    symmetric storage, the others get perturbed L blocks."""
    N, B = 9, 3
    d = synth.gen_numpy(n, N, seed=43, batch=B, dtype=np.float64)
    L, D, R = (np.array(x) for x in synth.unpack_bt(n, N, d["S"]))
    L[1, 4] *= 1.25
    L[2, 1:] += 0.01 * np.arange(n)[None, :, None]
    S = synth.pack_bt(L, D, R).astype(dtype)
    Lq, Dq, Rq = (np.array(x, dtype=np.float64) for x in synth.unpack_bt(n, N, S))
    want = synth.pack_bt(*synth.stair_pinv_blocks(Lq, Dq, Rq)).reshape(B, N, 3, n * n).copy()
    P = solver.form_pinv(n, N, B, dev(S), binding.PINV_STAIR)
    torch.cuda.synchronize()
    assert solver.check_symmetric(n, N, B, P).cpu().numpy().tolist() == [1, 0, 0]
    P = P.cpu().numpy().astype(np.float64).reshape(B, N, 3, n * n).copy()
    for arr in (P, want):
        arr[:, 0, 0] = 0
        arr[:, -1, 2] = 0
    assert relerr(P, want) < (1e-12 if dtype == np.float64 else 1e-5)


def test_full_size_config3_properties(solver):
    """BASELINE config 3 at full size (n=14, N=128, batch=1024, fp32): every problem converges in
    the generator's 9-10 iterations and the true residual ||gamma - S lambda|| / ||gamma|| is at
    the fp32 level; a second solve warm-started at the solution exits after one iteration."""
    n, N, B = 14, 128, 1024
    g = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=11)
    lam = torch.zeros_like(g["gamma"])
    iters, flags = solver.solve(n, N, B, g["S"], g["Pinv"], g["gamma"], lam, tol=1e-6, max_iter=50)
    torch.cuda.synchronize()
    it = iters.cpu().numpy()
    assert flags.sum().item() == 0 and it.min() >= 7 and it.max() <= 12
    res = g["gamma"] - solver.spmv(n, N, B, g["S"], lam)
    rel = res.norm(dim=1) / g["gamma"].norm(dim=1)
    assert float(rel.max()) < 5e-4
    iters2, _ = solver.solve(n, N, B, g["S"], g["Pinv"], g["gamma"], lam, tol=1e-6, max_iter=50)
    torch.cuda.synchronize()
    assert int(iters2.max()) == 1


# ------------------------------------------------------------------------------------- symmetric streaming
def _symmetrize_pinv(n, N, P):
    """Pinv with L_{k+1} := R_k^T exactly (the numpy stair blocks are each other's transpose only up to rounding)."""
    L, D, R = synth.unpack_bt(n, N, P)
    L = L.copy()
    L[..., 1:, :, :] = np.swapaxes(R[..., :-1, :, :], -1, -2)
    return synth.pack_bt(L, D, R)


def test_check_symmetric(solver):
    n, N, B = 14, 20, 5
    d = synth.gen_numpy(n, N, seed=61, batch=B, dtype=np.float32)
    S = d["S"].copy()
    assert solver.check_symmetric(n, N, B, dev(S)).cpu().numpy().tolist() == [1] * B   # generator: R_k = L_{k+1}^T exactly
    S[3, (7 * 3 + 2) * n * n + 5] += 1e-3                                                  # one element of R_7 of problem 3
    assert solver.check_symmetric(n, N, B, dev(S)).cpu().numpy().tolist() == [1, 1, 1, 0, 1]
    P = _symmetrize_pinv(n, N, d["Pinv"])
    assert solver.check_symmetric(n, N, B, dev(P)).cpu().numpy().tolist() == [1] * B


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_form_pinv_is_exactly_symmetric(solver, dtype):
    n, N, B = 14, 12, 4
    d = synth.gen_numpy(n, N, seed=62, batch=B, dtype=dtype)
    P = solver.form_pinv(n, N, B, dev(d["S"]), binding.PINV_STAIR)
    assert solver.check_symmetric(n, N, B, P).cpu().numpy().tolist() == [1] * B


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,N,B", [(14, 128, 8), (14, 100, 300), (14, 73, 2), (12, 40, 6), (16, 33, 5), (8, 50, 3)])
def test_symmetric_streaming_solve(solver, orc, dtype, n, N, B):
    """gbdpcg_set_symmetric: only [D|R] is read and L_{k+1} x_k is formed as R_k^T x_k.  On exactly
    symmetric storage the oracle (which reads L like the reference) must be matched as usual."""
    base = min(B, 8)
    d = synth.gen_numpy(n, N, seed=500 + n + N, batch=base, dtype=dtype)
    idx = np.arange(B) % base
    S = d["S"][idx]
    P = _symmetrize_pinv(n, N, d["Pinv"])[idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * np.arange(B))[:, None]).astype(dtype)
    solver.set_symmetric(1)      # assume: no device check, symmetric kernel for every problem
    try:
        out = gpu_solve(solver, n, N, B, S, P, g, tol=1e-6, max_iter=60, path=binding.PATH_FUSED)
    finally:
        solver.set_symmetric(2)  # back to the default (check on the device)
    ob = orc.pcg_batch(n, N, B, S, P, g, tol=1e-6, max_iter=60)
    assert np.array_equal(out["iters"], ob["iters"]) and not out["max_iter_exit"].any()
    tol = F64_TOL if dtype == np.float64 else F32_TOL
    for b in range(0, B, max(1, B // 8)):
        assert relerr(out["lambda_"][b], ob["lambda_"][b]) < tol


@pytest.mark.parametrize("N,B", [(128, 5), (127, 3), (101, 2), (74, 300), (73, 1)])
def test_symmetric_resident_kernel(solver, orc, N, B):
    """n = 14, fp32, 72 < N <= 128 with symmetric storage: both matrices stay on one CU for the whole
    solve (pcg_resident_sym.hip).  L blocks and R_{N-1} are NaN: only [D|R] of rows 0..N-2 and D_{N-1} may
    be read.  Non-zero initial guess; lambda, the final r and p and the iteration counts against the oracle,
    once to tolerance and once for a fixed iteration count."""
    n = 14
    base = min(B, 6)
    d = synth.gen_numpy(n, N, seed=900 + N, batch=base, dtype=np.float32)
    idx = np.arange(B) % base
    S = d["S"][idx].copy()
    P = _symmetrize_pinv(n, N, d["Pinv"])[idx].copy()
    g = (d["gamma"][idx] * (1.0 + 0.01 * np.arange(B))[:, None]).astype(np.float32)
    lam0 = np.stack([0.1 * synth.normals(950 + b, 0, n * N) for b in range(B)]).astype(np.float32)
    Sg, Pg = S.reshape(B, N, 3, n * n).copy(), P.reshape(B, N, 3, n * n).copy()
    for M in (Sg, Pg):
        M[:, :, 0, :] = np.nan      # every L block
        M[:, N - 1, 2, :] = np.nan  # R_{N-1}
    solver.set_symmetric(1)
    try:
        for tol, iters in ((1e-6, 60), (0.0, 7)):
            out = gpu_solve(solver, n, N, B, Sg.reshape(B, -1), Pg.reshape(B, -1), g, lam0=lam0, tol=tol,
                            max_iter=iters, path=binding.PATH_FUSED)
            ob = orc.pcg_batch(n, N, B, S, P, g, lambda0=lam0, tol=tol, max_iter=iters)
            assert np.array_equal(out["iters"], ob["iters"])
            assert np.array_equal(out["max_iter_exit"], ob["max_iter_exit"].astype(bool))
            for b in range(0, B, max(1, B // 8)):
                assert relerr(out["lambda_"][b], ob["lambda_"][b]) < F32_TOL
                if tol == 0.0:
                    scale = np.linalg.norm(g[b])
                    assert np.linalg.norm(out["r"][b] - ob["r"][b]) < 1e-5 * scale
                    assert np.linalg.norm(out["p"][b] - ob["p"][b]) < 1e-5 * scale
    finally:
        solver.set_symmetric(2)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("tol,max_iter", [(1e-6, 0), (1e-6, 1), (1e30, 5), (0.0, 2)])
def test_symmetric_resident_kernel_iteration_edges(solver, orc, mode, tol, max_iter):
    """max_iter = 0 / 1, an exit on the very first test, a fixed short run: iteration counts, the max-iter flag and
    the state left in lambda, r, p follow pcg.cuh:154,195,212 exactly as the oracle does."""
    n, N, B = 14, 100, 3
    d = synth.gen_numpy(n, N, seed=5, batch=B, dtype=np.float32)
    S, P, g = d["S"], _symmetrize_pinv(n, N, d["Pinv"]), d["gamma"]
    solver.set_symmetric(mode)
    try:
        out = gpu_solve(solver, n, N, B, S, P, g, tol=tol, max_iter=max_iter, path=binding.PATH_FUSED)
    finally:
        solver.set_symmetric(2)
    ob = orc.pcg_batch(n, N, B, S, P, g, tol=tol, max_iter=max_iter)
    assert np.array_equal(out["iters"], ob["iters"])
    assert np.array_equal(out["max_iter_exit"], ob["max_iter_exit"].astype(bool))
    scale = np.abs(g).max()
    for key in ("lambda_", "r", "p"):
        assert np.abs(out[key] - ob[key]).max() < 1e-5 * scale, key


def test_symmetric_resident_kernel_ill_conditioned(solver, orc):
    """Generator with a = 0.9 (kappa(S) of several hundred, 55 iterations): on the resident symmetric path the
    iteration counts stay within one of the fp32 oracle's and the distance to an fp64 solve of the same fp32 data
    is the oracle's own (iteration counts are order-sensitive on hard systems, so no exact equality here)."""
    n, N, B = 14, 128, 3
    d = synth.gen_numpy(n, N, seed=77, batch=B, dtype=np.float64, a=0.9)
    S32, g32 = d["S"].astype(np.float32), d["gamma"].astype(np.float32)
    P = solver.form_pinv(n, N, B, dev(S32), binding.PINV_STAIR)
    Ph = P.cpu().numpy()
    truth = orc.pcg_batch(n, N, B, S32.astype(np.float64), Ph.astype(np.float64), g32.astype(np.float64), tol=1e-14,
                          max_iter=500)
    o32 = orc.pcg_batch(n, N, B, S32, Ph, g32, tol=1e-6, max_iter=300)
    out = gpu_solve(solver, n, N, B, S32, Ph, g32, tol=1e-6, max_iter=300, path=binding.PATH_FUSED)
    assert not out["max_iter_exit"].any()
    # Why no equality here: at kappa ~ 800 the exit iteration depends on the summation order -- the oracle's own four
    # order variants (FMA contraction on / off x tree / sequential reduce, oracle/pcg_oracle.h) need not agree with each
    # other.  The GPU count has to lie in the band those variants span, widened by one on each side.
    band = np.stack([orc.pcg_batch(n, N, B, S32, Ph, g32, tol=1e-6, max_iter=300, flags=f)["iters"].astype(np.int64)
                     for f in range(4)])
    gi = out["iters"].astype(np.int64)
    assert (gi >= band.min(axis=0) - 1).all() and (gi <= band.max(axis=0) + 1).all(), (gi, band)
    assert np.abs(gi - o32["iters"].astype(np.int64)).max() <= 1 + (band.max(axis=0) - band.min(axis=0)).max()
    for b in range(B):
        e_gpu, e_orc = relerr(out["lambda_"][b], truth["lambda_"][b]), relerr(o32["lambda_"][b], truth["lambda_"][b])
        assert e_gpu < 1.5 * e_orc + 1e-6


def test_symmetric_resident_kernel_unaligned(solver, orc):
    """Matrices that are only 8-byte aligned (a view 2 floats into a buffer) take the resident kernel's
    direct-load form instead of the coalesced 16-byte one; same answers."""
    n, N, B = 14, 96, 3
    d = synth.gen_numpy(n, N, seed=990, batch=B, dtype=np.float32)
    S, P, g = d["S"], _symmetrize_pinv(n, N, d["Pinv"]), d["gamma"]
    ob = orc.pcg_batch(n, N, B, S, P, g, tol=1e-6, max_iter=60)

    def shifted(a):
        buf = torch.zeros(a.size + 2, dtype=torch.float32, device="cuda")
        v = buf[2:]
        v.copy_(torch.from_numpy(a.reshape(-1)))
        assert v.data_ptr() % 16 == 8
        return v

    dS, dP, dg = shifted(S), shifted(P), dev(g)
    lam = torch.zeros_like(dg)
    solver.set_symmetric(1)
    solver.set_path(binding.PATH_FUSED)
    try:
        iters, flags = solver.solve(n, N, B, dS, dP, dg, lam, tol=1e-6, max_iter=60)
        torch.cuda.synchronize()
    finally:
        solver.set_symmetric(2)
        solver.set_path(binding.PATH_AUTO)
    assert np.array_equal(iters.cpu().numpy().astype(np.int64), ob["iters"]) and not flags.cpu().numpy().any()
    lam = lam.cpu().numpy().reshape(B, -1)
    for b in range(B):
        assert relerr(lam[b], ob["lambda_"][b]) < F32_TOL


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mode", [0, 2])
def test_symmetric_auto_mixed_batch(solver, orc, dtype, mode):
    """Default mode 2: the symmetry relation is tested per problem on the device; a batch that mixes
    exactly symmetric problems with ones whose L blocks were perturbed (so that L_{k+1} != R_k^T, still a
    valid general block-tridiagonal input) must match the oracle on every problem -- symmetric ones through
    the [D|R] kernel, the others through the general one.  Mode 0 (never) is the control."""
    n, N, B = 14, 100, 300
    d = synth.gen_numpy(n, N, seed=77, batch=8, dtype=dtype)
    idx = np.arange(B) % 8
    S = d["S"][idx].copy()
    P = _symmetrize_pinv(n, N, d["Pinv"])[idx].copy()
    g = (d["gamma"][idx] * (1.0 + 0.01 * np.arange(B))[:, None]).astype(dtype)
    # break the relation in every third problem: scale one L block of S, one of Pinv in others
    for b in range(0, B, 3):
        S[b, (5 * 3 + 0) * n * n:(5 * 3 + 1) * n * n] *= 1.001
    for b in range(1, B, 7):
        P[b, (9 * 3 + 0) * n * n:(9 * 3 + 1) * n * n] *= 0.999
    flags = (solver.check_symmetric(n, N, B, dev(S)) & solver.check_symmetric(n, N, B, dev(P))).cpu().numpy()
    assert 0 < flags.sum() < B
    solver.set_symmetric(mode)
    try:
        out = gpu_solve(solver, n, N, B, S, P, g, tol=1e-6, max_iter=60, path=binding.PATH_FUSED)
    finally:
        solver.set_symmetric(2)
    ob = orc.pcg_batch(n, N, B, S, P, g, tol=1e-6, max_iter=60, nthreads=8)
    assert np.array_equal(out["iters"], ob["iters"]) and not out["max_iter_exit"].any()
    tol = F64_TOL if dtype == np.float64 else F32_TOL
    for b in range(B):
        assert relerr(out["lambda_"][b], ob["lambda_"][b]) < tol, (b, flags[b])


def test_symmetric_auto_in_graph(solver, orc):
    """The check + two-kernel dispatch of mode 2 is capturable: graph replays equal the eager solve."""
    n, N, B = 14, 90, 260
    d = synth.gen_numpy(n, N, seed=78, batch=4, dtype=np.float32)
    idx = np.arange(B) % 4
    S, P, g = d["S"][idx].copy(), _symmetrize_pinv(n, N, d["Pinv"])[idx].copy(), d["gamma"][idx].copy()
    S[1::2, (3 * 3) * n * n:(3 * 3 + 1) * n * n] *= 1.01   # odd problems: not symmetric
    eager = gpu_solve(solver, n, N, B, S, P, g, tol=1e-6, max_iter=40, path=binding.PATH_FUSED)
    dS, dP, dg = dev(S), dev(P), dev(g)
    lam = torch.zeros_like(dg)
    iters = torch.zeros(B, dtype=torch.int32, device="cuda")
    flags = torch.zeros(B, dtype=torch.uint8, device="cuda")
    solver.set_path(binding.PATH_FUSED)
    gr = solver.graph_solve(n, N, B, dS, dP, dg, lam, None, None, 1e-6, 40, iters, flags)
    solver.set_path(binding.PATH_AUTO)
    for _ in range(2):
        lam.zero_()
        gr.launch()
        torch.cuda.synchronize()
        assert np.array_equal(lam.cpu().numpy().reshape(B, -1), eager["lambda_"])
        assert np.array_equal(iters.cpu().numpy(), eager["iters"])
    gr.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,N,B", [(14, 128, 8), (14, 1, 3), (14, 2, 3), (14, 37, 5), (14, 300, 2), (12, 40, 6), (16, 33, 5), (8, 50, 3)])
def test_symmetric_spmv(solver, orc, dtype, n, N, B):
    """gbdpcg_set_symmetric(1) on the standalone SpMV: [D|R] only, chunk seams included (N = 300 is
    split over several workgroups per problem).  L blocks are poisoned with NaN: they must not be read."""
    d = synth.gen_numpy(n, N, seed=600 + n + N, batch=B, dtype=dtype)
    x = np.stack([synth.normals(700 + b, 0, n * N) for b in range(B)]).astype(dtype)
    S = d["S"].copy()
    want = np.stack([orc.dense_from_bt(n, N, S[b]) @ x[b].astype(np.float64) for b in range(B)])
    Sp = S.reshape(B, N, 3, n * n).copy()
    Sp[:, :, 0, :] = np.nan                      # every L block
    solver.set_symmetric(1)
    try:
        y = solver.spmv(n, N, B, dev(Sp.reshape(B, -1)), dev(x))
        torch.cuda.synchronize()
    finally:
        solver.set_symmetric(2)
    y = y.cpu().numpy()
    tol = 1e-13 if dtype == np.float64 else F32_TOL
    for b in range(B):
        assert relerr(y[b], want[b]) < tol


@pytest.mark.parametrize("mode", [2, 0])
def test_full_config3_batch_against_oracle(solver, orc, mode):
    """BASELINE config 3 in full -- n=14, N=128, fp32, all 1024 problems (seeds 1234+i) -- against the
    fp32 oracle problem by problem: equal iteration counts and lambda within 1e-6 norm-wise, through the
    default path (device symmetry check + [D|R] streaming; Pinv symmetrised exactly) and with mode 0."""
    n, N, B = 14, 128, 1024
    d = synth.gen_numpy(n, N, seed=1234, batch=B, dtype=np.float32)
    P = _symmetrize_pinv(n, N, d["Pinv"])
    if mode == 2:
        fl = (solver.check_symmetric(n, N, B, dev(d["S"])) & solver.check_symmetric(n, N, B, dev(P))).cpu().numpy()
        assert fl.all()
    solver.set_symmetric(mode)
    try:
        out = gpu_solve(solver, n, N, B, d["S"], P, d["gamma"], tol=1e-6, max_iter=50)
    finally:
        solver.set_symmetric(2)
    ob = orc.pcg_batch(n, N, B, d["S"], P, d["gamma"], tol=1e-6, max_iter=50, nthreads=8)
    assert np.array_equal(out["iters"], ob["iters"]) and not out["max_iter_exit"].any()
    num = np.linalg.norm(out["lambda_"].astype(np.float64) - ob["lambda_"], axis=1)
    den = np.linalg.norm(ob["lambda_"].astype(np.float64), axis=1)
    assert (num / den).max() < F32_TOL


# ------------------------------------------------------------------------------------- randomized dispatch sweep
def _sweep_cases(count=48, seed=20261004):
    # GBDPCG_SWEEP="count,seed": a longer hunt from the command line; the default is what CI runs
    if os.environ.get("GBDPCG_SWEEP"):
        count, seed = (int(x) for x in os.environ["GBDPCG_SWEEP"].split(","))
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(count):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 14, 14, 16, 18, 24, 25, 36, 40]))
        N = int(rng.choice([1, 2, 3, 9, 31, 64, 72, 73, 80, 81, 100, 128, 129, 168, 200, 256]))
        if n * n * N > 400000:
            N = max(1, 400000 // (n * n))
        B = int(rng.choice([1, 2, 3, 5, 9]))
        dtype = [np.float32, np.float64][int(rng.integers(2))]
        pinv = ["stair", "jacobi", None][int(rng.integers(3))]
        mode = int(rng.choice([0, 1, 2, 2]))
        fixed = bool(rng.integers(2)) and N > 2  # iterating past the exact solve of a tiny system is 0 / 0 or not by rounding luck
        cases.append((i, n, N, B, dtype, pinv, mode, fixed))
    # batches beyond the CU count: persistent workgroups, several rounds of the resident kernels, the symmetric
    # streaming kernels (which are only taken from 256 problems on)
    for j in range(max(8, count // 6)):
        n = int(rng.choice([4, 6, 8, 12, 14, 14, 14, 16]))
        N = int(rng.choice([5, 20, 64, 73, 100, 128]))
        B = int(rng.integers(256, 700))
        dtype = [np.float32, np.float64][int(rng.integers(2))]
        pinv = ["stair", "jacobi"][int(rng.integers(2))]
        cases.append((count + j, n, N, B, dtype, pinv, int(rng.choice([0, 1, 2, 2])), bool(rng.integers(2))))
    return cases


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: f"{c[0]}-n{c[1]}-N{c[2]}-B{c[3]}-{np.dtype(c[4]).name}-{c[5]}-m{c[6]}-{'fix' if c[7] else 'tol'}")
def test_randomized_dispatch_sweep(solver, orc, case):
    """Seeded random walk over block size, horizon, batch, precision, preconditioner kind (formed ON THE DEVICE, so
    the stair is exactly symmetric and every symmetric mode is legal), symmetric mode and exit rule, path AUTO:
    whichever kernel family the dispatch lands on must reproduce the oracle run on the same S and the same Pinv."""
    _, n, N, B, dtype, pinv, mode, fixed = case
    base = min(B, 6)  # large batches repeat six generated systems with rescaled right-hand sides
    d = synth.gen_numpy(n, N, seed=7000 + case[0], batch=base, dtype=dtype)
    idx = np.arange(B) % base
    S = np.ascontiguousarray(d["S"][idx])
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // base))[:, None]).astype(dtype)
    dS = dev(S)
    P_h = None
    if pinv is not None:
        kind = binding.PINV_STAIR if pinv == "stair" else binding.PINV_BLOCK_JACOBI
        P_h = solver.form_pinv(n, N, B, dS, kind).cpu().numpy()
    tol, max_iter = (0.0, 6) if fixed else (1e-6, 200)
    solver.set_symmetric(mode)
    try:
        out = gpu_solve(solver, n, N, B, S, P_h, g, tol=tol, max_iter=max_iter)
    finally:
        solver.set_symmetric(2)
    ob = orc.pcg_batch(n, N, B, S, P_h, g, tol=tol, max_iter=max_iter)
    bound = (F64_TOL if dtype == np.float64 else F32_TOL) * (1 if pinv is not None else 4)  # no preconditioner: kappa 25-30
    assert np.array_equal(out["max_iter_exit"], ob["max_iter_exit"].astype(bool))
    # Without a preconditioner the runs to tolerance take 13-30 iterations (finite-termination regime for the tiny
    # systems): where rounding puts the exit differs between two summation orders by up to a few iterations and the
    # iterates drift apart by far more than an ulp (order-sensitive), so those cases only have to
    # exit within three iterations of the oracle, at a point as close to the dense solve as the exit rule implies.
    long_run = pinv is None and not fixed
    if long_run:
        # ... and the band is the oracle's own: its four summation-order variants (FMA on / off x tree / sequential
        # reduce) bracket the exit iteration; the GPU count has to lie within that band widened by two
        band = np.stack([orc.pcg_batch(n, N, B, S, P_h, g, tol=tol, max_iter=max_iter, flags=f)["iters"].astype(np.int64)
                         for f in range(4)])
        gi = out["iters"].astype(np.int64)
        assert (gi >= band.min(axis=0) - 2).all() and (gi <= band.max(axis=0) + 2).all(), (gi, band)
        assert np.abs(gi - ob["iters"].astype(np.int64)).max() <= 3
    else:
        assert np.array_equal(out["iters"], ob["iters"].astype(np.int64))
    for b in range(B):
        if not np.isfinite(ob["lambda_"][b]).all():
            # fixed iteration count past an exact solve (N = 1 with an exact preconditioner): 0 / 0 in pcg.cuh:169
            assert not np.isfinite(out["lambda_"][b]).all()
        elif long_run:
            truth = np.linalg.solve(orc.dense_from_bt(n, N, S[b]), g[b].astype(np.float64))
            e_gpu, e_orc = relerr(out["lambda_"][b], truth), relerr(ob["lambda_"][b], truth)
            assert e_gpu < max(2 * e_orc + 10 * bound, 2e-3), (b, e_gpu, e_orc)  # exit rule: |r.r| < 1e-6, kappa < 30
        else:
            assert relerr(out["lambda_"][b], ob["lambda_"][b]) < bound, (b, out["iters"][b])


def _aux_cases(count=36, seed=424242):
    if os.environ.get("GBDPCG_SWEEP"):
        count, seed = (int(x) for x in os.environ["GBDPCG_SWEEP"].split(","))
        seed += 1000
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([1, 2, 3, 4, 6, 7, 8, 12, 13, 14, 16, 18, 24, 25, 36, 40]))
        N = int(rng.choice([1, 2, 3, 14, 15, 16, 17, 30, 31, 46, 64]))
        B = int(rng.choice([1, 2, 3, 7, 40]))
        if n * n * N * B > 600000:
            B = max(1, 600000 // (n * n * N))
        out.append((i, n, N, B, [np.float32, np.float64][int(rng.integers(2))], bool(rng.integers(2))))
    return out


@pytest.mark.parametrize("case", _aux_cases(), ids=lambda c: f"{c[0]}-n{c[1]}-N{c[2]}-B{c[3]}-{np.dtype(c[4]).name}-{'asym' if c[5] else 'sym'}")
def test_randomized_pinv_spmv_check(solver, orc, case):
    """The operators either side of the solve over random shapes: stair formation (symmetric S: mirrored, exactly
    symmetric output; perturbed S: both slots evaluated) against the host construction in fp64, the symmetry test's
    per-problem verdicts, and the block-tridiagonal product against the dense one."""
    i, n, N, B, dtype, asym = case
    d = synth.gen_numpy(n, N, seed=9000 + i, batch=B, dtype=np.float64)
    L, D, R = (np.array(x) for x in synth.unpack_bt(n, N, d["S"]))
    touched = np.zeros(B, bool)
    if asym and N > 1:
        rng = np.random.default_rng(i)
        for b in range(B):
            if rng.integers(2):
                k = int(rng.integers(1, N))
                L[b, k, int(rng.integers(n)), int(rng.integers(n))] *= 1.0 + 2.0 ** -12
                touched[b] = True
    S = synth.pack_bt(L, D, R).astype(dtype)
    dS = dev(S)
    assert solver.check_symmetric(n, N, B, dS).cpu().numpy().astype(bool).tolist() == (~touched).tolist()
    P = solver.form_pinv(n, N, B, dS, binding.PINV_STAIR)
    torch.cuda.synchronize()
    # symmetric S => exactly symmetric Pinv; the converse can fail by rounding (a last-bits change of a tiny element
    # of S may vanish in the products), so nothing is asserted about the verdict of a perturbed problem's Pinv
    p_sym = solver.check_symmetric(n, N, B, P).cpu().numpy().astype(bool)
    assert p_sym[~touched].all()
    Lq, Dq, Rq = (np.array(x, dtype=np.float64) for x in synth.unpack_bt(n, N, S))
    want = synth.pack_bt(*synth.stair_pinv_blocks(Lq, Dq, Rq)).reshape(B, N, 3, n * n).copy()
    got = P.cpu().numpy().astype(np.float64).reshape(B, N, 3, n * n).copy()
    for arr in (got, want):
        arr[:, 0, 0] = 0
        arr[:, -1, 2] = 0
    assert relerr(got, want) < (1e-11 if dtype == np.float64 else 3e-5)
    x = np.stack([synth.normals(77 + b, i, n * N) for b in range(B)]).astype(dtype)
    y = solver.spmv(n, N, B, dS, dev(x)).cpu().numpy()
    for b in range(min(B, 4)):
        A = orc.dense_from_bt(n, N, S[b])
        assert relerr(y[b], A @ x[b].astype(np.float64)) < (1e-13 if dtype == np.float64 else F32_TOL)
