"""Pins the CPU oracle (oracle/pcg_oracle.c) -- CPU only.

The reference holds no expected outputs, so the pins are:
  * the input system the reference's examples hold (examples/pcg_solve.cu:14-25) solved
    densely in fp64 -- the mathematical known answer -- and the iteration counts
    SURVEY.md section 8c records for it (6 with Pinv = I, 3 with the symmetric stair);
  * dense fp64 solves / products of generated systems at every BASELINE.json shape;
  * the committed golden files (regression: generator + oracle unchanged).
Last-bit parity with the CUDA reference stays unpinned (GLASS summation order unknown).
"""
import hashlib
import os

import numpy as np
import pytest

from gbd_pcg_amd import synth

SHAPES = [(2, 3), (3, 5), (7, 2), (14, 1), (14, 64), (14, 128), (36, 256)]

# lambda* of SURVEY.md section 4 (dense fp64 solve of the README system)
README_LAMBDA = np.array([-303.702986086, -46.415939681, -315.176302632,
                          -14.898309418, -298.790861920, 13.503782688])


def relerr(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b)


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def test_readme_known_answer(orc):
    n, N, S, gamma = orc.readme_system()
    A = orc.dense_from_bt(n, N, S)
    lam = np.linalg.solve(A, gamma)
    assert np.allclose(lam, README_LAMBDA, rtol=0, atol=1e-8)
    # S is symmetric negative definite here; CG is sign-agnostic
    assert np.allclose(A, A.T) and np.all(np.linalg.eigvalsh(A) < 0)

    out = orc.pcg(n, N, S, None, gamma, tol=1e-6, max_iter=25)
    assert out["iters"] == 6 and not out["max_iter_exit"]
    assert relerr(out["lambda_"], lam) < 1e-12

    L, D, R = synth.unpack_bt(n, N, S)
    Pinv = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    out = orc.pcg(n, N, S, Pinv, gamma, tol=1e-6, max_iter=25)
    assert out["iters"] == 3
    assert relerr(out["lambda_"], lam) < 1e-12


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_readme_fp32_orders(orc, flags):
    """fp32 on the kappa~1562 README system: every summation order converges to the same
    answer at ~1e-5; the iteration count with Pinv = I is order-sensitive (8 or 9), with
    the stair preconditioner it is 3 for every order."""
    n, N, S, gamma = orc.readme_system()
    L, D, R = synth.unpack_bt(n, N, S)
    Pinv = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    o = orc.pcg(n, N, S.astype(np.float32), None, gamma.astype(np.float32), flags=flags)
    assert o["iters"] in (8, 9) and relerr(o["lambda_"], README_LAMBDA) < 2e-4
    o = orc.pcg(n, N, S.astype(np.float32), Pinv.astype(np.float32), gamma.astype(np.float32),
                flags=flags)
    assert o["iters"] == 3 and relerr(o["lambda_"], README_LAMBDA) < 5e-5


@pytest.mark.parametrize("n,N", SHAPES)
def test_spmv_vs_dense(orc, n, N):
    d = synth.gen_numpy(n, N, seed=99, a=0.5)
    A = orc.dense_from_bt(n, N, d["S"][0])
    x = synth.normals(5, 0, n * N)
    for flags in (0, orc.DEFAULT):
        y = orc.spmv(n, N, d["S"][0], x, flags=flags)
        assert relerr(y, A @ x) < 1e-14
    y32 = orc.spmv(n, N, d["S"][0].astype(np.float32), x.astype(np.float32))
    assert relerr(y32, A @ x) < 5e-7


def test_spmv_ignores_unused_corner_blocks(orc):
    """L_0 and R_{N-1} are never read (pcg.cuh:105-106, utils.cuh:58-75): NaNs there are harmless."""
    n, N = 3, 4
    d = synth.gen_numpy(n, N, seed=3)
    S = d["S"][0].copy()
    x = synth.normals(6, 0, n * N)
    y0 = orc.spmv(n, N, S, x)
    S[: n * n] = np.nan
    S[-n * n:] = np.nan
    assert np.array_equal(orc.spmv(n, N, S, x), y0)


@pytest.mark.parametrize("n,N", SHAPES)
def test_pcg_vs_dense_solve(orc, n, N):
    d = synth.gen_numpy(n, N, seed=1234, a=0.5)
    S, P, g = d["S"][0], d["Pinv"][0], d["gamma"][0]
    A = orc.dense_from_bt(n, N, S)
    assert np.allclose(A, A.T) and np.linalg.eigvalsh(A)[0] > 0
    lam = np.linalg.solve(A, g)
    # converge hard: the oracle is then the dense answer
    o = orc.pcg(n, N, S, P, g, tol=1e-26, max_iter=200)
    assert not o["max_iter_exit"] and relerr(o["lambda_"], lam) < 1e-11
    o = orc.pcg(n, N, S, None, g, tol=1e-26, max_iter=400)
    assert not o["max_iter_exit"] and relerr(o["lambda_"], lam) < 1e-11
    # residual state left in r: r = gamma - S lambda
    assert np.linalg.norm(o["r"] - (g - A @ o["lambda_"])) < 1e-10 * np.linalg.norm(g)


def test_exit_semantics(orc):
    """pcg.cuh:101,154,195,212: iters = i+1 on convergence at loop index i; on running out,
    iters = max_iter and max_iter_exit = true; max_iter = 0 leaves lambda untouched."""
    d = synth.gen_numpy(14, 16, seed=7)
    S, P, g = d["S"][0], d["Pinv"][0], d["gamma"][0]
    full = orc.pcg(14, 16, S, P, g, tol=1e-6, max_iter=50, trace=True)
    k = full["iters"]
    assert not full["max_iter_exit"] and abs(full["eta"][k]) < 1e-6 <= abs(full["eta"][k - 1])
    cut = orc.pcg(14, 16, S, P, g, tol=1e-6, max_iter=k - 1)
    assert cut["iters"] == k - 1 and cut["max_iter_exit"]
    fixed = orc.pcg(14, 16, S, P, g, tol=0.0, max_iter=25)       # abs(eta) < 0 never true
    assert fixed["iters"] == 25 and fixed["max_iter_exit"]
    lam0 = synth.normals(8, 0, 14 * 16)
    none = orc.pcg(14, 16, S, P, g, lambda0=lam0, max_iter=0)
    assert none["iters"] == 0 and none["max_iter_exit"] and np.array_equal(none["lambda_"], lam0)


def test_warm_start(orc):
    """lambda is in/out (README.md:50, pcg.cuh:119,215): starting at the solution exits in 1."""
    d = synth.gen_numpy(14, 16, seed=11)
    S, P, g = d["S"][0], d["Pinv"][0], d["gamma"][0]
    lam = np.linalg.solve(orc.dense_from_bt(14, 16, S), g)
    o = orc.pcg(14, 16, S, P, g, lambda0=lam, tol=1e-6)
    assert o["iters"] == 1 and relerr(o["lambda_"], lam) < 1e-12


def test_batch_matches_single(orc):
    d = synth.gen_numpy(14, 8, seed=21, batch=5, dtype=np.float32)
    ob = orc.pcg_batch(14, 8, 5, d["S"], d["Pinv"], d["gamma"], nthreads=3)
    for i in range(5):
        o = orc.pcg(14, 8, d["S"][i], d["Pinv"][i], d["gamma"][i])
        assert np.array_equal(o["lambda_"], ob["lambda_"][i]) and o["iters"] == ob["iters"][i]


@pytest.mark.parametrize("name", ["readme", "gen_2x3", "gen_3x5", "gen_7x2", "gen_14x1",
                                  "gen_14x64", "gen_14x128", "gen_36x256"])
def test_golden_files(orc, golden_dir, name):
    """Committed fixtures still describe what generator + oracle produce (and the dense answer)."""
    G = np.load(os.path.join(golden_dir, name + ".npz"))
    n, N = int(G["n"]), int(G["N"])
    if name == "readme":
        _, _, S, g = orc.readme_system()
        L, D, R = synth.unpack_bt(n, N, S)
        P = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    else:
        d = synth.gen_numpy(n, N, seed=int(G["seed"]), a=float(G["a"]))
        S, P, g = d["S"][0], d["Pinv"][0], d["gamma"][0]
        assert checksum(S, P, g) == str(G["checksum"])
    for tag, Pm in (("stair", P), ("ident", None)):
        o = orc.pcg(n, N, S, Pm, g, tol=1e-6, max_iter=100, trace=True)
        assert o["iters"] == int(G[f"iters_f64_{tag}"])
        assert relerr(o["lambda_"], G[f"lambda_f64_{tag}"]) < 1e-13
        assert relerr(o["lambda_"], G["lambda_dense"]) < 1e-3
        o32 = orc.pcg(n, N, S.astype(np.float32), None if Pm is None else Pm.astype(np.float32),
                      g.astype(np.float32), tol=1e-6, max_iter=100)
        assert o32["iters"] == int(G[f"iters_f32_{tag}"])
    assert relerr(orc.spmv(n, N, S, G["x"]), G["y_dense"]) < 1e-14


def test_survey_iteration_counts(golden_dir):
    """9 / 9 / 10 iterations at (14,64) / (14,128) / (36,256) with the
    stair preconditioner at tol 1e-6 (the survey's random draws differ; the counts are a
    property of the generator's conditioning)."""
    for name, want in (("gen_14x64", 9), ("gen_14x128", 9), ("gen_36x256", 10)):
        G = np.load(os.path.join(golden_dir, name + ".npz"))
        assert int(G["iters_f64_stair"]) == want and int(G["iters_f32_stair"]) == want


def test_oracle_under_address_and_ub_sanitizers():
    """The C restatement built with -fsanitize=address,undefined (CPU only; the GPU pool has no sanitizer runs) and
    driven by oracle/oracle_selftest.c: the reference's example system under every flag combination (6 iterations,
    SURVEY.md section 8c), ragged small shapes with NaN-poisoned L_0 / R_{N-1} in exact-size buffers, both
    precisions, 1 and 3 OpenMP threads.  Any out-of-bounds access, signed overflow or misaligned access aborts
    the binary."""
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", odir, "asan"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(odir, "oracle_selftest_asan")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="3"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "oracle selftest ok" in out.stdout and "FAIL" not in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
