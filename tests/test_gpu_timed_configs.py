"""The configurations bench.py TIMES, compared with the CPU oracle exactly as they are timed.

bench.py measures fixed-count solves -- exit_tol = 0, max_iter = 25, so that |eta'| < 0 never holds and every problem
runs all 25 iterations (/root/reference/include/pcg.cuh:195; the count and flag written at :212) -- replayed from a
hipGraph, on problems Gen(n, N, 1234 + i, 0.5) built on the device (synth.gen_torch_seeded) with the symmetric-stair
Phi^-1 formed on the device (gbdpcg_form_pinv).  The parity tests elsewhere run to tolerance (9-10 iterations) or stop
at 7 fixed iterations; here the timed workload itself -- fifteen iterations past convergence included, where alpha =
eta / (p.Sp) is a quotient of rounding-sized numbers -- is checked against the oracle run on the same matrices (the
device-formed Phi^-1 is read back, so both sides multiply the same bits):

    iters == 25 and max_iter_exit == 1 for EVERY problem of the batch,
    lambda within 1e-6 (fp32) / 1e-10 (fp64) norm-wise for >= 64 problems spread over the batch,
    r and p (the state pcg.cuh:175,205 leave behind) within 2e-5 of gamma's scale,
    and the true residual ||gamma - S lambda|| / ||gamma|| of every problem, from gbdpcg_spmv, below a stated bound.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from gbd_pcg_amd import binding, synth  # noqa: E402

pytestmark = pytest.mark.gpu

BASE_SEED, MAX_ITER = 1234, 25          # bench.py: BASE_SEED, MAX_ITER


@pytest.fixture(scope="module")
def solver():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    s = binding.Solver(0)
    yield s
    s.close()


def workload(solver, n, N, B, dtype):
    """What bench.py builds for a config: problems 0 .. B-1 of Gen(n, N, 1234 + i, 0.5), stair Phi^-1 formed on the device."""
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", dtype, seed=BASE_SEED)
    S, gamma = g["S"], g["gamma"]
    del g
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    return S, P, gamma


def timed_solve(solver, n, N, B, S, P, gamma, mode=2, replays=3):
    """One bench step: lambda = 0, then the captured solve (exit_tol 0, 25 iterations) replayed -- several times, as the
    bench does, so that what is compared is a REPLAY of the graph and not its first run."""
    lam = torch.zeros_like(gamma)
    r, p = torch.full_like(gamma, float("nan")), torch.full_like(gamma, float("nan"))
    it = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    fl = torch.full((B,), 7, dtype=torch.uint8, device="cuda")
    solver.set_symmetric(mode)
    try:
        gr = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, 0.0, MAX_ITER, it, fl)
    finally:
        solver.set_symmetric(2)
    for _ in range(replays):
        lam.zero_()
        gr.launch()
    torch.cuda.synchronize()
    gr.close()
    return lam, r, p, it, fl


def true_residual(solver, n, N, B, S, gamma, lam):
    y = solver.spmv(n, N, B, S, lam)
    torch.cuda.synchronize()
    num = (gamma.double() - y.double()).reshape(B, -1).norm(dim=1)
    return (num / gamma.double().reshape(B, -1).norm(dim=1)).cpu().numpy()


def compare(orc, n, N, B, S, P, gamma, lam, r, p, it, fl, idx, ltol):
    it = it.cpu().numpy().astype(np.int64)
    fl = fl.cpu().numpy().astype(np.int64)
    assert (it == MAX_ITER).all(), (it.min(), it.max())
    assert (fl == 1).all(), np.unique(fl)
    assert bool(torch.isfinite(lam).all()) and bool(torch.isfinite(r).all()) and bool(torch.isfinite(p).all())
    sel = torch.as_tensor(idx, device="cuda")
    hS, hP, hg = (t.reshape(B, -1)[sel].cpu().numpy() for t in (S, P, gamma))
    ob = orc.pcg_batch(n, N, len(idx), hS, hP, hg, tol=0.0, max_iter=MAX_ITER, nthreads=8)
    assert (ob["iters"] == MAX_ITER).all() and ob["max_iter_exit"].all()
    hl, hr, hp = (t.reshape(B, -1)[sel].cpu().numpy().astype(np.float64) for t in (lam, r, p))
    worst = 0.0
    for j in range(len(idx)):
        err = np.linalg.norm(hl[j] - ob["lambda_"][j]) / np.linalg.norm(ob["lambda_"][j])
        worst = max(worst, err)
        assert err < ltol, (idx[j], err)
        scale = np.abs(hg[j]).max()
        assert np.abs(hr[j] - ob["r"][j]).max() < 2e-5 * scale, idx[j]
        assert np.abs(hp[j] - ob["p"][j]).max() < 2e-5 * scale, idx[j]
    return worst


@pytest.mark.parametrize("mode", [2, 1, 0])
def test_headline_config3_as_timed(solver, orc, mode):
    """BASELINE configs[2] exactly as bench.py times it (value, roofline and general_kernel blocks: symmetric modes 2, 1
    and 0): 1024 problems, 64 of them (every 16th) against the oracle, all of them by their true residual."""
    n, N, B = 14, 128, 1024
    S, P, gamma = workload(solver, n, N, B, torch.float32)
    lam, r, p, it, fl = timed_solve(solver, n, N, B, S, P, gamma, mode=mode)
    compare(orc, n, N, B, S, P, gamma, lam, r, p, it, fl, list(range(5, B, 16)), 1e-6)
    res = true_residual(solver, n, N, B, S, gamma, lam)
    assert res.max() < 2e-6, res.max()          # fp32 rounding floor of r = gamma - S lambda at kappa(S) ~ 27: measured 4.9e-7


def test_config2_as_timed(solver, orc):
    """BASELINE configs[1] (n = 14, N = 64, fp32, one problem: the register-resident kernel) at the bench's fixed 25."""
    n, N, B = 14, 64, 1
    S, P, gamma = workload(solver, n, N, B, torch.float32)
    assert solver.choose_path(4, n, N, B) == binding.PATH_FUSED
    lam, r, p, it, fl = timed_solve(solver, n, N, B, S, P, gamma)
    compare(orc, n, N, B, S, P, gamma, lam, r, p, it, fl, [0], 1e-6)
    assert true_residual(solver, n, N, B, S, gamma, lam).max() < 2e-6


def test_config4_as_timed(solver, orc):
    """BASELINE configs[3] (n = 36, N = 256, fp64, one problem: the persistent launch) at the bench's fixed 25."""
    n, N, B = 36, 256, 1
    S, P, gamma = workload(solver, n, N, B, torch.float64)
    assert solver.choose_path(8, n, N, B) == binding.PATH_PERSISTENT
    lam, r, p, it, fl = timed_solve(solver, n, N, B, S, P, gamma)
    compare(orc, n, N, B, S, P, gamma, lam, r, p, it, fl, [0], 1e-10)
    assert true_residual(solver, n, N, B, S, gamma, lam).max() < 1e-12   # measured 1.4e-13


def test_config5_slice_as_timed(solver, orc):
    """BASELINE configs[4]'s 8192 problems on one GPU (the bench's C5_on_one_gpu block) at fixed 25: every problem by
    count, flag and true residual, 64 of them (every 128th) against the oracle."""
    n, N, B = 14, 128, 8192
    S, P, gamma = workload(solver, n, N, B, torch.float32)
    lam, r, p, it, fl = timed_solve(solver, n, N, B, S, P, gamma, replays=2)
    compare(orc, n, N, B, S, P, gamma, lam, r, p, it, fl, list(range(77, B, 128)), 1e-6)
    assert true_residual(solver, n, N, B, S, gamma, lam).max() < 2e-6
