"""The C++ drop-in surface (include/gbdpcg.hpp: solvePCG<T> x3, pcg_solve<T>, pcg_config<T>, csr_t<T>,
pcgSharedMemSize<T>, checkPcgOccupancy<T>) driven from compiled host code, the way MPCGPU would
call it, checked against the CPU oracle on the reference's example system."""
import os
import subprocess

import numpy as np
import pytest

from gbd_pcg_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "gbd-pcg_amd", "examples")


def run(name):
    exe = os.path.join(EX, name)
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() makes it"
    return subprocess.run([exe], capture_output=True, text=True, timeout=120, check=True).stdout


def parse(out):
    rec = {}
    for line in out.splitlines():
        if " iters=" in line:
            tag, rest = line.split(" iters=")
            it, lam = rest.split(" lambda=")
            rec[tag] = (int(it), np.array([float(v) for v in lam.split()]))
    return rec


def test_selftest_against_oracle(orc):
    out = run("api_selftest")
    rec = parse(out)
    n, N, S, gamma = orc.readme_system()
    lam_star = np.linalg.solve(orc.dense_from_bt(n, N, S), gamma)
    L, D, R = synth.unpack_bt(n, N, S)
    P = synth.pack_bt(*synth.stair_pinv_blocks(L, D, R))
    assert "f64 smem=400 occupancy=1" in out and "f32 smem=200 occupancy=1" in out   # pcg.cuh:13-20

    o_id = orc.pcg(n, N, S, None, gamma, tol=1e-6, max_iter=25)
    o_st = orc.pcg(n, N, S, P, gamma, tol=1e-6, max_iter=25)
    for tag in ("host_ident", "pcg_solve", "device_ident", "csr_ident"):
        it, lam = rec[f"f64 {tag}"]
        assert it == o_id["iters"] == 6
        assert np.linalg.norm(lam - o_id["lambda_"]) / np.linalg.norm(o_id["lambda_"]) < 1e-10
        assert np.linalg.norm(lam - lam_star) / np.linalg.norm(lam_star) < 1e-10
    it, lam = rec["f64 host_stair"]
    assert it == o_st["iters"] == 3 and np.linalg.norm(lam - o_st["lambda_"]) / np.linalg.norm(lam_star) < 1e-10
    it, res = rec["f64 device_resid"]
    assert np.linalg.norm(res - o_id["r"]) < 1e-9 * np.linalg.norm(gamma)

    # device-formed stair Pinv, read back: the oracle runs on EXACTLY the preconditioner the device solve used, in the
    # device's precision, so the iteration counts are compared for equality (L_0 / R_{N-1} of the read-back are never
    # used, pcg.cuh:105-106)
    pinv = {}
    for line in out.splitlines():
        if " device_stair_pinv=" in line:
            prec, vals = line.split(" device_stair_pinv=")
            pinv[prec] = np.array([float(v) for v in vals.split()])
    for prec, dt, tol in (("f64", np.float64, 1e-10), ("f32", np.float32, 2e-5)):
        P_dev = np.nan_to_num(pinv[prec]).astype(dt)
        o_dev = orc.pcg(n, N, S.astype(dt), P_dev, gamma.astype(dt), tol=1e-6, max_iter=25)
        it, lam = rec[f"{prec} device_stair"]
        assert it == o_dev["iters"], (prec, it, o_dev["iters"])
        assert np.linalg.norm(lam - o_dev["lambda_"]) / np.linalg.norm(o_dev["lambda_"]) < tol
        it_h, lam_h = rec[f"{prec} host_stair"]      # the host overload forms the same Pinv: same count, same answer
        assert it_h == it and np.array_equal(lam_h, lam)

    # fp32, Pinv = I: kappa(S) ~ 1562 and no preconditioner -- the exit iteration depends on the summation order
    # (the reference compiled for the host takes 9, a numpy restatement 8; the oracle's own
    # FMA / tree-order variants bracket it), so the GPU count has to be one some summation order produces
    S32, g32 = S.astype(np.float32), gamma.astype(np.float32)
    counts = {int(orc.pcg(n, N, S32, None, g32, tol=1e-6, max_iter=25, flags=f)["iters"]) for f in range(4)} | {8, 9}
    for tag in ("host_ident", "pcg_solve", "device_ident", "csr_ident"):
        it, lam = rec[f"f32 {tag}"]
        assert it in counts and np.linalg.norm(lam - lam_star) / np.linalg.norm(lam_star) < 2e-4


def test_selftest_kkt_wrappers():
    """formSchur<T> / recoverPrimal<T> / kktStep<T> of include/gbdpcg.hpp on the 2-state, 1-control, 3-knot problem written out
    in examples/api_selftest.cpp: multipliers and primal step against numpy.linalg.solve of the whole KKT system."""
    from oracle import schur_oracle as so
    out = run("api_selftest")
    rec = parse(out)
    G = [2, 0.5, 0.5, 1, 3, 1.5, 0.2, 0.2, 2, 1, 1, 0, 0, 4]
    C = [1, 0.1, -0.2, 0.9, 0.5, 1, 0.8, 0, 0.3, 1.1, 0, 0.7]
    g = [1, -1, 0.5, 0.3, 0.2, -0.4, -0.6, 0.9]
    c = [0.5, -0.25, 0.1, 0.2, -0.3, 0.05]
    z_star, lam_star = so.dense_kkt_solve(2, 1, 3, G, C, g, c)
    for prec, tol in (("f64", 1e-11), ("f32", 2e-5)):
        it, lam = rec[f"{prec} kkt_step"]
        assert 1 <= it < 50 and np.linalg.norm(lam - lam_star) < tol * np.linalg.norm(lam_star)
        z = [np.array([float(v) for v in ln.split("=")[1].split()]) for ln in out.splitlines() if ln.startswith(f"{prec} kkt_z=")][0]
        assert np.linalg.norm(z - z_star) < tol * np.linalg.norm(z_star)
        assert f"{prec} kkt_wrappers_agree=1" in out


@pytest.mark.parametrize("exe", ["pcg_solve", "pcg_solve_dp"])
def test_example_drivers_print_like_the_reference(exe):
    """examples/pcg_solve.cu:36-41 prints 'GBD-PCG returned in <res> iters.' then 'Lambda: ' and six values."""
    out = run(exe).splitlines()
    assert out[0].startswith("GBD-PCG returned in ") and out[0].endswith(" iters.")
    assert out[1].strip() == "Lambda:"
    lam = np.array([float(v) for v in out[2].split()])
    want = np.array([-303.702986086, -46.415939681, -315.176302632, -14.898309418, -298.790861920, 13.503782688])
    assert lam.shape == (6,) and np.linalg.norm(lam - want) / np.linalg.norm(want) < (2e-4 if exe == "pcg_solve" else 1e-5)


def test_batched_control_loop_example():
    """examples/mpc_batch_loop.cpp: the C ABI the way a batched MPC pipeline would use it (Pinv formed on the device,
    one graph replayed per control step).  The program returns 0 only if the true residual of the solve is small."""
    exe = os.path.join(EX, "mpc_batch_loop")
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() makes it"
    out = subprocess.run([exe, "48", "100", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "step 1:" in out.stdout and "||gamma - S lambda||" in out.stdout


def test_kkt_step_loop_example():
    """examples/kkt_step_loop.cpp: one graph replay per SQP step on the KKT blocks themselves (gbdpcg_graph_create_kkt_step_f32),
    warm-started; the program checks both KKT residuals of what comes back in fp64 on the host and returns 0 only if they are
    small and no problem ran out of iterations.  Horizons that are and are not a multiple of 4."""
    exe = os.path.join(EX, "kkt_step_loop")
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() makes it"
    for args in (["40", "64", "3"], ["5", "19", "2"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "step 1:" in out.stdout and "stationarity" in out.stdout and out.stdout.strip().endswith("ok")
