"""BASELINE configs[4] (stateSize 14, knotPoints 128, fp32, batch 8192 over 8 GPUs) as far as ONE GPU can show it:
the whole 8192-problem batch on one device, two handles in one process (what a one-process-eight-handles host
driver does per device), the bench launcher with two ranks, and the prefetch A/B build of the headline kernel."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from gbd_pcg_amd import binding, sharding, synth  # noqa: E402

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


def test_config5_whole_batch_on_one_gpu(solver, orc):
    """8192 problems Gen(14, 128, 1234 + i, 0.5), stair Pinv formed on the device, default
    path, converge to 1e-6: every problem takes 7..12 iterations (9 on this generator, section 8c), the true
    residual is small, a warm restart from the solution exits after one iteration, and a 64-problem subset from
    both ends and the middle of the batch matches the oracle run on the same device-formed Pinv problem by problem
    (equal iteration counts, lambda within 1e-6 norm-wise)."""
    n, N, B = 14, 128, 8192
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32, seed=1234)
    S, gamma = g["S"], g["gamma"]
    del g
    # the inputs are the seeds-1234+i problems: spot-check against the canonical numpy generator
    for i in (0, 1023, 1024, 8191):
        ref = synth.gen_numpy(n, N, seed=1234 + i, batch=1, dtype=np.float32)
        assert relerr(S[i].cpu().numpy(), ref["S"][0]) < 1e-6 and relerr(gamma[i].cpu().numpy(), ref["gamma"][0]) < 1e-6
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    r, p = torch.empty_like(gamma), torch.empty_like(gamma)
    iters, flags = solver.solve(n, N, B, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)
    torch.cuda.synchronize()
    it = iters.cpu().numpy()
    assert flags.sum().item() == 0 and it.min() >= 7 and it.max() <= 12, (it.min(), it.max())
    res = gamma - solver.spmv(n, N, B, S, lam)
    rel = (res.norm(dim=1) / gamma.norm(dim=1)).max().item()
    assert rel < 2e-3, rel       # exit rule is |r . Pinv r| < 1e-6 on O(40)-norm right-hand sides
    # the r the solver leaves behind is that residual (recursively updated: equal up to fp32 rounding of gamma's scale)
    assert ((r - res).norm(dim=1) / gamma.norm(dim=1)).max().item() < 1e-5
    # subset against the oracle
    idx = np.r_[0:24, 4090:4106, 8168:8192]
    ob = orc.pcg_batch(n, N, len(idx), S[idx].cpu().numpy(), P[idx].cpu().numpy(), gamma[idx].cpu().numpy(),
                       tol=1e-6, max_iter=25, nthreads=8)
    assert np.array_equal(it[idx], ob["iters"].astype(np.int64))
    lam_h = lam[idx].cpu().numpy()
    for k in range(len(idx)):
        assert relerr(lam_h[k], ob["lambda_"][k]) < 1e-6, (idx[k], it[idx[k]])
    # warm restart: lambda is already the solution
    iters2, flags2 = solver.solve(n, N, B, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)
    torch.cuda.synchronize()
    assert int(iters2.max()) == 1 and int(iters2.min()) == 1 and flags2.sum().item() == 0
    # rank g of 8 would own exactly problems [1024 g, 1024 (g+1))
    assert [sharding.shard_range(B, r_, 8) for r_ in (0, 7)] == [(0, 1024), (7168, 8192)]


def test_two_handles_in_one_process(solver, orc):
    """A second handle on the same device, and two host threads that each own a handle and a stream, run the
    headline shape (n = 14, N = 128, fp32: the 160 KB-LDS resident kernel, whose dynamic-LDS opt-in used to be cached
    in a process-wide static) and agree bit for bit with the first handle and with the oracle's iteration counts."""
    n, N, B = 14, 128, 6
    d = synth.gen_numpy(n, N, seed=4242, batch=B, dtype=np.float32)
    S, g = torch.from_numpy(d["S"]).cuda(), torch.from_numpy(d["gamma"]).cuda()
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    ob = orc.pcg_batch(n, N, B, d["S"], P.cpu().numpy(), d["gamma"], tol=1e-6, max_iter=50)

    def solve_with(s, stream=None):
        lam = torch.zeros_like(g)
        it, fl = s.solve(n, N, B, S, P, g, lam, tol=1e-6, max_iter=50, stream=stream)
        (stream or torch.cuda.current_stream()).synchronize()
        return lam.cpu().numpy(), it.cpu().numpy()

    lam0, it0 = solve_with(solver)
    assert np.array_equal(it0, ob["iters"].astype(np.int32))
    second = binding.Solver(0)
    try:
        lam1, it1 = solve_with(second)
        assert np.array_equal(lam0, lam1) and np.array_equal(it0, it1)
    finally:
        second.close()

    results, errors = {}, []

    def worker(tag):
        try:
            torch.cuda.set_device(0)
            s = binding.Solver(0)
            st = torch.cuda.Stream()
            for _ in range(3):
                results[tag] = solve_with(s, st)
            s.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((tag, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for tag in ("a", "b"):
        assert np.array_equal(results[tag][0], lam0) and np.array_equal(results[tag][1], it0)


def test_calls_restore_the_callers_device(solver):
    """Entry points make the handle's device current for the call only (include/gbdpcg.h, lifetime section)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0
    before = cur.value
    n, N, B = 14, 8, 2
    d = synth.gen_numpy(n, N, seed=7, batch=B, dtype=np.float32)
    S, g = torch.from_numpy(d["S"]).cuda(), torch.from_numpy(d["gamma"]).cuda()
    solver.solve(n, N, B, S, None, g, torch.zeros_like(g), tol=1e-6, max_iter=5)
    torch.cuda.synchronize()
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == before


_AB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from gbd_pcg_amd import binding, synth
s = binding.Solver(0)
out = {}
for tag, N, B, tol, mi in (("c3", 128, 520, 1e-6, 25), ("fix", 128, 300, 0.0, 7), ("odd", 101, 261, 1e-6, 25)):
    d = synth.gen_numpy(14, N, seed=900, batch=4, dtype=np.float32)
    rep = (B + 3) // 4
    S = torch.from_numpy(np.tile(d["S"], (rep, 1))[:B].copy()).cuda()
    g = torch.from_numpy((np.tile(d["gamma"], (rep, 1))[:B] * (1 + 0.001 * np.arange(B))[:, None]).astype(np.float32)).cuda()
    P = s.form_pinv(14, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(g); r = torch.empty_like(g); p = torch.empty_like(g)
    it, fl = s.solve(14, N, B, S, P, g, lam, r, p, tol=tol, max_iter=mi)
    torch.cuda.synchronize()
    out[tag + "_lam"] = lam.cpu().numpy(); out[tag + "_r"] = r.cpu().numpy(); out[tag + "_p"] = p.cpu().numpy()
    out[tag + "_it"] = it.cpu().numpy()
np.savez(sys.argv[2], **out)
"""


def test_prefetch_off_build_is_bit_identical(tmp_path):
    """The resident kernel pulls the next problem's [D|R] lines towards the Infinity Cache with LDS-DMA loads that
    hipcc's vmcnt bookkeeping does not see (pcg_resident_sym.hip, symres_touch).  They have no register destination
    and land in a dump area nobody reads, so they must not change a single bit: the same solves through the library
    built with -DGBDPCG_RS_PREFETCH=0 (csrc/variants/libgbdpcg_nopf.so) give identical lambda, r, p and iteration
    counts.  Batches > 256 problems so that every workgroup prefetches at least once (and one ends mid-round)."""
    variant = os.path.join(ROOT, "gbd-pcg_amd", "csrc", "variants", "libgbdpcg_nopf.so")
    assert os.path.exists(variant), "variants/libgbdpcg_nopf.so missing: __graft_entry__.build() makes it"
    outs = []
    for tag, lib in (("shipped", None), ("nopf", variant)):
        env = dict(os.environ)
        env.pop("GBDPCG_LIB", None)
        if lib:
            env["GBDPCG_LIB"] = lib
        path = str(tmp_path / f"{tag}.npz")
        res = subprocess.run([sys.executable, "-c", _AB_SCRIPT, ROOT, path], env=env, capture_output=True, text=True,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        outs.append(np.load(path))
    a, b = outs
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["c3_it"].min() >= 7 and np.isfinite(a["c3_lam"]).all()


def test_bench_launcher_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` with no torch.distributed environment: the launcher starts two fresh ranks
    before any GPU call; here both land on cuda:0 and aggregate over gloo (GBDPCG_BENCH_BACKEND=gloo -- the
    rehearsal form for a one-GPU box; on a multi-GPU node the same command uses RCCL, one rank per GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GBDPCG_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2048 and rec["scaling"] == "weak"
    assert rec["value"] > 1e6 and rec["unit"] == "iter/s"
    assert 0.0 < rec["roofline"]["frac"] <= 1.0
    assert "cpu_baseline" not in rec      # rank 0 at N = 1 only


def test_multi_device_cpp_driver():
    """examples/multi_gpu_batch.cpp: one host thread + handle + graph per device and RCCL all-reduces of
    {problems solved, iteration sum} / max elapsed.  This box has one device, so one rank -- the program returns 0
    only if every problem converged with a small true residual and the all-reduced count equals the batch."""
    exe = os.path.join(ROOT, "gbd-pcg_amd", "examples", "multi_gpu_batch")
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() makes it"
    res = subprocess.run([exe, "600", "128", "2", "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "RCCL ranks = 1" in res.stdout and "problems solved = 600 of 600" in res.stdout


def test_bench_line_contract():
    """`python bench.py` (the driver's command, N = 1): one JSON line with the metric of BASELINE.json, a roofline block
    whose fraction is a fraction, the SpMV block over the 1.23 GB rotation, the per-config block and the CPU baseline."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3"],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert rec["metric"] == base["metric"] and rec["unit"] == "iter/s" and rec["n_gpus"] == 1
    assert rec["steps"] == 10 and rec["warmup"] == 3 and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    assert rec["dtype"] == "f32" and rec["data"] == "synthetic" and "workload" in rec["config"]
    assert abs(rec["value"] - 1024 * 25 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    rf = rec["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu") and 0.0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["all_problems_symmetric"] is True and rf["replays"] >= 100
    sp = rec["spmv"]
    assert sp["bound"] == "hbm" and 0.5 < sp["frac"] <= 1.0 and "1.23 GB" in sp["rotation"]
    assert set(rec["configs"]) == {"C2", "C4", "C5_on_one_gpu"} and rec["configs"]["C4"]["path"] == "persistent"
    assert rec["configs"]["C4"]["iters_converged"] == 10.0 and rec["configs"]["C2"]["iters_converged"] == 9.0
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    ms = rec["mpc_step"]   # the steps either side of the solve at the headline batch shape
    assert "error" not in ms and ms["S_symmetric_in_storage"] is True and ms["all_converged_and_finite"] is True
    for k in ("form_schur", "recover_primal"):
        assert ms[k]["bound"] == "hbm" and 0.1 < ms[k]["frac"] <= 1.0 and abs(ms[k]["frac"] - ms[k]["achieved"] / ms[k]["peak"]) < 1e-12
    assert ms["us_per_step_of_1024_problems"] < ms["form_schur"]["us"] + ms["form_pinv_solve"]["us"] + ms["recover_primal"]["us"] + 50.0
