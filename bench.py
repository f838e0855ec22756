#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native block-tridiagonal PCG.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process is only a LAUNCHER -- it starts N fresh child
ranks (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before making any GPU call, waits for
them and prints rank 0's JSON line.  Under torch.distributed.run (WORLD_SIZE set) it is one rank; WORLD_SIZE
must then equal --gpus.

Workload (BASELINE.json configs[2], the one the north-star target is quoted on): stateSize n = 14,
knotPoints N = 128, fp32, 1024 independent problems PER GPU (configs[4] is this shape at 8 GPUs: weak
scaling, no data-path collective).  The job's batch is problems 0 .. 1024*N-1 of SURVEY.md section 8d's
Gen(14, 128, 1234 + i, 0.5); rank g owns the contiguous slice sharding.shard_range gives it and builds it
on its own GPU.  Phi^-1 = symmetric stair formed on the device from S.

A "step" = one batched PCG solve with a fixed iteration count (exit_tol = 0, max_iter = 25: the test
|eta| < 0 never holds, /root/reference/include/pcg.cuh:195), lambda reset to 0 first, replayed from a
hipGraph.  value = problem-iterations per second over the whole job (all ranks' units / max elapsed).

Printed JSON line (rank 0) also carries
  roofline     : the dominant kernel.  pcg_resident_sym_kernel keeps [D|R] of S and Phi^-1 on the CU for the
                 whole solve, so its bound is fp32 vector issue, not HBM: achieved = flop/s against the
                 157.3 TFLOP/s fp32 peak (frac <= 1), with the HBM share of the bytes it really moves and the
                 section-8d algorithmic-bytes figure as a labelled "equivalent stream rate"
  spmv         : the standalone block-tridiagonal SpMV (the kernel the >= 70 % HBM target is quoted on), timed
                 over 4 rotating matrices (1.23 GB > the 256 MiB Infinity Cache): an HBM statement
  configs      : driver-timed numbers for the other single-GPU BASELINE configs (C2, C4, C5's batch on one GPU)
  mpc_step     : the steps either side of the solve at the headline batch shape: KKT blocks -> S, gamma
                 (HBM-bound, with its fraction) -> Phi^-1 + converged PCG -> primal step (HBM-bound, with its fraction)
  cpu_baseline : the CPU oracle (oracle/pcg_oracle.c, a port -- the reference has no CPU path)
                 timed on this host's cores on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEIL_GBPS = 6290.0  # measured float4-copy ceiling, same table
HBM_COLD_READ_GBPS = 6185.0  # best pure-read kernel (non-temporal loads) on Infinity-Cache-cold data (profiles/r01_bw_probe_cold.txt)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same table: peak FP32 (vector) = the dense fp32 MFMA peak

N_STATE, N_KNOTS, BATCH_PER_GPU, MAX_ITER = 14, 128, 1024, 25
BASE_SEED = 1234


def profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel_prefix):
    """HBM-side bytes per launch from the committed PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of this
    same benchmark, gfx950 x2 read correction calibrated on known-size reads).  None when no profile matches."""
    for f in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        prof = profile_json(f)
        if not prof:
            continue
        for name, rec in prof.get("kernels", {}).items():
            if name.startswith(kernel_prefix):
                return rec["traffic_bytes_per_launch"]
    return None


def schur_traffic(kernel):
    """HBM-side bytes per launch of a schur.hip kernel from the committed PMC passes of tools/schur_run.py."""
    prof = profile_json("r03_schur_kernels.json") or profile_json("r02_schur_kernels.json")
    rec = (prof or {}).get("kernels", {}).get(kernel)
    return rec.get("traffic_bytes_per_launch") if rec else None


def pmc_valu(kernel_prefix):
    """VALU-issue utilisation of a kernel from the committed SQ-counter pass (profiles/r02_pmc_sq.json, written by
    gbd-pcg_amd/tools/pmc_sq_json.py): SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES per SIMD, LDS bank-conflict share."""
    prof = profile_json("r03_pmc_sq.json") or profile_json("r02_pmc_sq.json")
    if not prof:
        return None
    for name, rec in prof.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            return rec
    return None


def pcg_flops_per_launch(n, N, batch, iters):
    """2 flops per stored element of the touched blocks, two products per iteration and two in the prologue
    (S lambda, Pinv r): (2 iters + 2) * 2 (3N-2) n^2 per problem (SURVEY.md section 8d flop count)."""
    return batch * (2 * iters + 2) * 2 * (3 * N - 2) * n * n


STREAMING_CHILD = r"""
import sys, json, torch
sys.path.insert(0, sys.argv[1])
from gbd_pcg_amd import binding, synth
n, N, B, iters = (int(v) for v in sys.argv[2:6])
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma); it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
s.set_symmetric(0)
gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0, iters, it, fl)
for _ in range(5): lam.zero_(); gr.launch()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
torch.cuda.synchronize()
for a, b in evs: a.record(); gr.launch(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in evs)
print(json.dumps({"kernel_ms": t[15]}))
"""


def time_streaming_general(n, N, B, iters):
    """pcg_fused_kernel<float,14,2,8,false> (both matrices streamed every iteration) on the same workload, in a child
    process with the cluster path switched off; None if the child fails."""
    import subprocess
    env = dict(os.environ, GBDPCG_NO_CLUSTER="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        o = subprocess.run([sys.executable, "-c", STREAMING_CHILD, os.path.dirname(os.path.abspath(__file__)), str(n), str(N),
                            str(B), str(iters)], env=env, capture_output=True, text=True, timeout=180)
        line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
        ms = json.loads(line[-1])["kernel_ms"]
    except Exception:
        return None
    by = pcg_bytes_per_launch(n, N, B, iters, 4)
    return {"kernel": "pcg_fused_kernel<float,14,2,8,false> (GBDPCG_NO_CLUSTER=1)", "kernel_ms": ms,
            "problem_iters_per_sec_one_gpu": B * iters / (ms * 1e-3), "algorithmic_GBps": by / (ms * 1e-3) / 1e9,
            "note": "algorithmic bytes / time; the in-flight set is re-read every iteration, so part is served by the 256 MiB "
                    "Infinity Cache: not an HBM-only statement"}


def pcg_bytes_per_launch(n, N, batch, iters, s):
    """Algorithmic HBM bytes of one solve by SURVEY.md section 8d: S and Pinv streamed once per iteration
    (2 (3N-2) n^2 s) + the prologue's one pass over each + vectors (gamma, lambda in; lambda, r, p out)."""
    mat = (3 * N - 2) * n * n * s
    return batch * ((2 * iters + 2) * mat + 5 * n * N * s)


def spmv_bytes_per_launch(n, N, batch, s):
    """((3N-2) n^2 + 2 n N) s per problem."""
    return batch * ((3 * N - 2) * n * n + 2 * n * N) * s


def host_cores():
    """CPU share of this process: cgroup quota if one is set, else the affinity mask."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(n, N, iters, budget_s=12.0):
    """Oracle (port) on the host cores, bounded sample of the same workload."""
    import numpy as np
    from gbd_pcg_amd import synth
    from oracle import oracle as orc
    cores = host_cores()
    probe = max(4 * cores, 32)
    d = synth.gen_numpy(n, N, seed=BASE_SEED, batch=probe, dtype=np.float32)
    t0 = time.perf_counter()
    orc.pcg_batch(n, N, probe, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=iters, nthreads=cores)
    t_probe = time.perf_counter() - t0
    reps = max(1, min(int(budget_s / max(t_probe, 1e-4)), 20000))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.pcg_batch(n, N, probe, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=iters, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": probe * reps * iters / dt, "unit": "iter/s", "cores": cores, "kind": "port",
            "sample": f"problems 0..{probe - 1} of the workload x {reps} repeats x {iters} iterations, n={n} N={N} fp32, "
                      f"OpenMP over problems, {dt:.1f} s"}


# ---------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torch.distributed environment
# ---------------------------------------------------------------------------------------------------------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks):
    """Start n_ranks fresh child processes of this script, one per GPU, and relay rank 0's output.  The parent
    makes no GPU call (it never imports torch): every rank initialises its own device in its own process.  Every
    rank's stderr is kept and, if the job fails, its tail is forwarded, so that a first run on a real multi-GPU node
    that dies says which rank died of what."""
    import tempfile
    port = free_port()
    procs, errs = [], []
    with tempfile.TemporaryFile() as out0:
        for rank in range(n_ranks):
            env = dict(os.environ)
            env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n_ranks),
                        "LOCAL_WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            errs.append(tempfile.TemporaryFile())
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if rank == 0 else subprocess.DEVNULL, stderr=errs[-1]))
        # wait for all ranks; a rank that dies takes the job down (the others would sit in the rendezvous or a barrier)
        failed, first_dead = False, None
        while any(p.poll() is None for p in procs):
            dead = [i for i, p in enumerate(procs) if p.returncode not in (None, 0)]
            if dead:
                failed = True
                if first_dead is None and len(dead) == 1:
                    first_dead = dead[0]          # seen dead while every other rank was still running
                for p in procs:
                    if p.poll() is None:
                        p.kill()              # exactly the children started above, by PID
            time.sleep(0.05)
        codes = [p.returncode for p in procs]
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
        failed = failed or any(codes)
        for rank, f in enumerate(errs):
            f.seek(0)
            text = f.read().decode(errors="replace")
            f.close()
            if failed and text.strip():
                mark = " (first to die)" if rank == first_dead else ""
                sys.stderr.write(f"---- bench.py rank {rank}, exit code {codes[rank]}{mark}: last lines of stderr ----\n")
                sys.stderr.write(text[-3000:].rstrip() + "\n")
            elif text.strip():
                sys.stderr.write(text)       # warnings of a healthy run pass through
    if failed:
        sys.stderr.write(f"bench.py launcher: rank exit codes {codes}\n")
        sys.exit(1)


# ---------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------

def run_dry(args, world, rank):
    """Host logic of the N > 1 path with no GPU (CPU tests, gloo): rendezvous, sharding, aggregation, the
    JSON line's bookkeeping.  Solves nothing and reports no metric value."""
    import torch
    import torch.distributed as dist
    from gbd_pcg_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if os.environ.get("GBDPCG_BENCH_DRY_FAIL_RANK") == str(rank):    # tests/test_distributed.py: what the launcher reports
        raise RuntimeError(f"injected failure on rank {rank} (dry run)")
    total = BATCH_PER_GPU * world
    lo, hi = sharding.shard_range(total, rank, world)
    spans = [None] * world
    if world > 1:
        dist.all_gather_object(spans, (lo, hi, os.getpid()))
        dist.barrier()
    else:
        spans = [(lo, hi, os.getpid())]
    elapsed, units = sharding.aggregate(1.0 + rank, float((hi - lo) * MAX_ITER * args.steps), device="cpu")
    if rank == 0:
        print(json.dumps({"metric": "PCG iterations/sec and GB/s on block-tridiag SpMV, stateSize×knotPoints",
                          "value": None, "dry_run": True, "n_gpus": dist.get_world_size() if world > 1 else 1,
                          "steps": args.steps, "warmup": args.warmup, "shards": [list(s[:2]) for s in spans],
                          "pids": [s[2] for s in spans], "seeds": f"{BASE_SEED} + problem index",
                          "max_elapsed": elapsed, "total_units": units}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


# Bounds on the true residual ||gamma - S lambda|| / ||gamma|| of a timed buffer (asserted, outside the timed region):
# fixed-count runs sit on the rounding floor of their precision (kappa(S) ~ 27: measured 5e-7 in fp32, 1.4e-13 in fp64 at n = 36, N = 256);
# runs to exit_tol = 1e-6 stop on |r.Pinv r| < 1e-6 (pcg.cuh:195), an ABSOLUTE test: ||r|| <~ 1e-3 against ||gamma|| ~ 40.
RESIDUAL_BOUND = {("f32", "fixed"): 2e-6, ("f64", "fixed"): 1e-12, ("f32", "converged"): 1e-3, ("f64", "converged"): 1e-3}


def verify_solve(solver, torch, n, N, B, S, gamma, lam, it, fl, kind, fixed_iters=None):
    """What the timed replays left in their buffers, checked through the library's own SpMV (general kernel: reads L, D
    and R): true residual of EVERY problem, iteration counts and exit flags.  Raises if the timed work was not a solve."""
    y = solver.spmv(n, N, B, S, lam)
    torch.cuda.synchronize()
    g = gamma.double().reshape(B, -1)
    res = ((g - y.double().reshape(B, -1)).norm(dim=1) / g.norm(dim=1))
    dt = "f32" if gamma.dtype == torch.float32 else "f64"
    # a fixed count too short to converge (the 5-iteration runs that price one iteration) has no residual to promise
    bound = RESIDUAL_BOUND[(dt, kind)] if kind != "fixed" or fixed_iters >= MAX_ITER else float("inf")
    itc, flc = it.to(torch.int64) & 0xffffffff, fl.to(torch.int64)
    rec = {"true_residual_max": float(res.max()), "true_residual_median": float(res.median()), "bound": bound,
           "problems": B, "iters_min": int(itc.min()), "iters_max": int(itc.max()),
           "max_iter_exit_values": sorted(int(v) for v in torch.unique(flc).cpu()),
           "finite": bool(torch.isfinite(lam).all()),
           "definition": "max over all problems of ||gamma - S lambda|| / ||gamma|| on the buffers the timed replays left, "
                         "S applied by gbdpcg_spmv (general kernel)"}
    ok = rec["finite"] and rec["true_residual_max"] < bound
    if kind == "fixed":      # exit_tol = 0: every problem runs out of iterations (pcg.cuh:195,212)
        ok = ok and rec["iters_min"] == rec["iters_max"] == fixed_iters and rec["max_iter_exit_values"] == [1]
    else:                    # to tolerance: every problem converged before max_iter
        ok = ok and rec["iters_min"] >= 1 and rec["max_iter_exit_values"] == [0]
    rec["ok"] = bool(ok)
    assert ok, f"timed buffers do not hold a solve: {rec}"
    return rec


def run_rank(args, world, rank, local_rank):
    import torch
    import torch.distributed as dist
    from gbd_pcg_amd import binding, sharding, synth

    distributed = world > 1
    # GBDPCG_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a one-GPU box (every rank on cuda:0,
    # aggregation on CPU tensors); the real multi-GPU run uses RCCL ("nccl") with one rank per GPU
    backend = os.environ.get("GBDPCG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        world = dist.get_world_size()   # the ranks the collective library actually sees
    agg_device = "cuda" if backend == "nccl" else "cpu"
    dev = torch.device("cuda", local_rank)

    n, N, iters = N_STATE, N_KNOTS, MAX_ITER
    total_batch = BATCH_PER_GPU * world
    lo, hi = sharding.shard_range(total_batch, rank, world)   # rank g owns problems [g B/G, (g+1) B/G)
    B = hi - lo
    solver = binding.Solver(local_rank)
    g = synth.gen_torch_seeded(n, N, lo, hi, dev, torch.float32, seed=BASE_SEED)
    S, gamma = g["S"], g["gamma"]
    del g
    # Phi^-1 = symmetric stair, formed on the device from S (gbdpcg_form_pinv: exactly symmetric storage)
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    r, p = torch.empty_like(gamma), torch.empty_like(gamma)
    it_out = torch.zeros(B, dtype=torch.int32, device=dev)
    fl_out = torch.zeros(B, dtype=torch.uint8, device=dev)
    graph = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, 0.0, iters, it_out, fl_out)
    stream = torch.cuda.current_stream()

    def new_events(k):
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]

    def step(ev=None):
        lam.zero_()
        if ev:
            ev[0].record(stream)
        graph.launch(stream)
        if ev:
            ev[1].record(stream)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP-event pairs around every EV_EVERY-th step only: an event pair costs the stream ~7 us (gbd-pcg_amd/tools/step_overhead.py:
    # 0.4226 ms per step with a pair around every step, 0.4155 ms with none), and the events are instrumentation, not work
    EV_EVERY = 5
    events = new_events((args.steps + EV_EVERY - 1) // EV_EVERY)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k // EV_EVERY] if k % EV_EVERY == 0 else None)
    fence()
    elapsed = time.perf_counter() - t0
    # RCCL over xGMI: throughput aggregation only (max elapsed, total problem-iterations)
    elapsed, total_units = sharding.aggregate(elapsed, float(B * iters * args.steps), device=agg_device)
    verified = verify_solve(solver, torch, n, N, B, S, gamma, lam, it_out, fl_out, "fixed", iters)   # outside the timed region
    step_ms = median([a.elapsed_time(b) for a, b in events])   # check kernel + both PCG launches of one replay

    def time_graph(gr, reps, warm=5, reset=None):
        """Median (and minimum) per-replay HIP-event time of a solve graph on the launch stream."""
        reset = lam if reset is None else reset
        for _ in range(warm):
            reset.zero_()
            gr.launch(stream)
        evs = new_events(reps)
        torch.cuda.synchronize()
        for e0, e1 in evs:
            reset.zero_()
            e0.record(stream)
            gr.launch(stream)
            e1.record(stream)
        torch.cuda.synchronize()
        ts = [e0.elapsed_time(e1) for e0, e1 in evs]
        return median(ts), min(ts)

    def time_mode(mode, reps):
        solver.set_symmetric(mode)
        gr = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, 0.0, iters, it_out, fl_out)
        solver.set_symmetric(2)
        med, best = time_graph(gr, reps)
        ver = verify_solve(solver, torch, n, N, B, S, gamma, lam, it_out, fl_out, "fixed", iters)
        gr.close()
        return med, best, ver

    all_symmetric = (int(solver.check_symmetric(n, N, B, S).min()) == 1
                     and int(solver.check_symmetric(n, N, B, P).min()) == 1)
    REPS = max(100, args.steps)
    sym_ms, sym_best, sym_ver = time_mode(1, REPS)   # the dominant kernel alone (caller asserts symmetry: no check launch)

    def time_converged(mode, reps):
        """The same kernel run to exit_tol = 1e-6 (what an MPC loop runs: 9 iterations on this generator)."""
        solver.set_symmetric(mode)
        gr = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, 1e-6, iters, it_out, fl_out)
        solver.set_symmetric(2)
        med, best = time_graph(gr, reps)
        ver = verify_solve(solver, torch, n, N, B, S, gamma, lam, it_out, fl_out, "converged")
        mean_it = float(it_out.float().mean())
        gr.close()
        return med, best, ver, mean_it

    # (--no-converged: the kernel-stats pass of tools/profile_round.sh leaves these dispatches out, so that the AverageNs rocprofv3
    # reports for the dominant kernel is the average of the fixed-25 launches roofline.kernel_ms is measured on)
    if args.no_converged:
        conv_ms = conv_best = dflt_ms = conv_it = 1.0   # (unused: the block below says "skipped")
        conv_ver = dflt_ver = None
    else:
        conv_ms, conv_best, conv_ver, conv_it = time_converged(1, REPS)
        dflt_ms, _, dflt_ver, _ = time_converged(2, REPS)
    flops = pcg_flops_per_launch(n, N, B, iters)
    pcg_bytes = pcg_bytes_per_launch(n, N, B, iters, 4)
    resident_bytes = B * (2 * (2 * N - 1) * n * n + 5 * n * N) * 4   # [D|R] of both matrices once per solve + vectors
    tflops = flops / (sym_ms * 1e-3) / 1e12
    sq = pmc_valu("pcg_resident_sym_kernel")
    out = None
    if rank == 0:
        # Phi^-1 operand accounting of the resident kernel ("LDS hit rate for the preconditioner", north star):
        # [D|R] of Phi^-1 is consumed (iters + 1) times per solve and fetched from HBM once; on chip, block-rows
        # k1 = 2j+1 live in LDS whole and P0_LDS_QUADS = 2 of the 14 pieces of block-rows k0 = 2j as well.
        lds_share = (14 + 2) / 28.0
        out = {
            "metric": "PCG iterations/sec and GB/s on block-tridiag SpMV, stateSize×knotPoints",
            "value": total_units / elapsed,
            "unit": "iter/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if distributed else 1,
            "collective_backend": (dist.get_backend() if distributed else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: stateSize=14 knotPoints=128 fp32 batch=1024 per GPU "
                                   "(configs[4] at 8 GPUs), 25 fixed PCG iterations per step (exit_tol=0), problems "
                                   "Gen(14,128,1234+i,0.5), symmetric-stair Pinv formed on the device, hipGraph replay",
                       "stateSize": n, "knotPoints": N, "batch_per_gpu": BATCH_PER_GPU, "global_batch": total_batch,
                       "pcg_iters_per_step": iters,
                       "path": "default (symmetric mode 2): device check of L_{k+1} == R_k^T, then [D|R] of S and Pinv "
                               "stay resident on the CU (registers + LDS) for the whole solve",
                       "graph_ms_per_step_median": step_ms,
                       "sharding": f"problems [g*{BATCH_PER_GPU}, (g+1)*{BATCH_PER_GPU}) on rank g of {world}, seeds 1234+i, "
                                   "no data-path collective; RCCL all_reduce of (max elapsed, sum units) only"},
            "solves_per_sec": total_batch * args.steps / elapsed,
            "verified": verified,
            "value_definition": "problem-iterations per second (25 PCG iterations x 1024 problems per GPU per step), default path",
            "roofline": {
                "bound": "valu",
                "kernel": "pcg_resident_sym_kernel<14,true> (symmetric S and Pinv resident on the CU; timed alone, "
                          "gbdpcg_set_symmetric(1))",
                "achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
                "traffic": pmc_traffic("pcg_resident_sym_kernel"),
                "flops_per_launch": flops,
                "kernel_ms": sym_ms, "kernel_ms_min": sym_best, "replays": REPS, "statistic": "median",
                "valu_issue": sq,
                "hbm_share": {"bytes_moved_per_launch": resident_bytes,
                              "achieved_GBps": resident_bytes / (sym_ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": resident_bytes / (sym_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "equivalent_stream_rate": {"algorithmic_bytes_per_launch": pcg_bytes,
                                           "GBps": pcg_bytes / (sym_ms * 1e-3) / 1e9,
                                           "note": "SURVEY 8d bytes (S and Pinv in full, once per iteration) / kernel time: "
                                                   "what a streaming kernel would have to sustain to match; not an HBM rate"},
                "all_problems_symmetric": all_symmetric,
                "verified": sym_ver,
                "converged": {"skipped": "--no-converged"} if args.no_converged else {
                              "bound": "per-CU ingest (fabric), then valu", "exit_tol": 1e-6, "iters_mean": conv_it,
                              "kernel_ms": conv_ms, "kernel_ms_min": conv_best,
                              "default_path_ms_incl_symmetry_check": dflt_ms,
                              "solves_per_sec_one_gpu": B / (conv_ms * 1e-3),
                              "achieved": resident_bytes / (conv_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": resident_bytes / (conv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "bytes_moved_per_launch": resident_bytes,
                              "tflops": pcg_flops_per_launch(n, N, B, conv_it) / (conv_ms * 1e-3) / 1e12,
                              "frac_of_fp32_peak": pcg_flops_per_launch(n, N, B, conv_it) / (conv_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                              "verified": conv_ver, "verified_default_path": dflt_ver,
                              "note": "the number an MPC loop sees: the solve stops at |r.Pinv r| < 1e-6; per round of 256 problems "
                                      "about 17 us are tile loads at the per-CU ingest rate (401 KB at ~33 GB/s per CU, "
                                      "profiles/r03_resident_stamps.txt) + prologue + write-back, the rest 2.2 us per iteration"},
                "note": "peak = fp32 vector peak (= dense fp32 MFMA peak) of MI355X_MICROARCH.md; the matrices are read once "
                        "per solve, so the kernel is bound by VALU issue and on-chip latency, not by HBM"},
            "pinv_onchip": {"hit_rate": iters / (iters + 1.0), "from_lds": lds_share, "from_registers": 1.0 - lds_share,
                            "definition": "Phi^-1 bytes consumed by the (iters+1) products of a solve that are served from "
                                          "the CU (LDS or registers) / bytes consumed; from_lds + from_registers split the "
                                          "on-chip part (kernel geometry, pcg_resident_sym.hip)"},
        }

    if world == 1:
        try:
            # General (not bit-symmetric) storage: gbdpcg_set_symmetric(0) makes every problem take the path a caller-formed
            # Phi^-1 takes.  Since round 2 that is the cluster kernel (both matrices register-resident over two CUs per
            # problem, two cross-CU hand-offs per iteration); the kernel that streams both matrices every iteration is timed in
            # a child process with GBDPCG_NO_CLUSTER=1 (the switch is read once per process).
            gen_ms, gen_best, gen_ver = time_mode(0, 60)
            gen_tflops = flops / (gen_ms * 1e-3) / 1e12
            full_bytes = B * (2 * 3 * N * n * n + 5 * n * N) * 4      # [L|D|R] of both matrices once per solve + vectors
            clusters = 256 // 2
            rounds = -(-B // clusters)
            out["general_kernel"] = {
                "kernel": "pcg_cluster_kernel<14,2,true> (gbdpcg_set_symmetric(0): general storage, two CUs per problem)",
                "bound": "latency", "achieved": gen_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": gen_tflops / FP32_VECTOR_PEAK_TFLOPS, "kernel_ms": gen_ms, "kernel_ms_min": gen_best,
                "traffic": pmc_traffic("pcg_cluster_kernel"),
                "problem_iters_per_sec_one_gpu": B * iters / (gen_ms * 1e-3),
                "us_per_iteration_of_a_cluster": gen_ms * 1e3 / rounds / (iters + 1),
                "handoff_floor_us_per_iteration": 2 * 0.5,
                "hbm_share": {"bytes_moved_per_launch": full_bytes, "achieved_GBps": full_bytes / (gen_ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": full_bytes / (gen_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "streaming_kernel": time_streaming_general(n, N, B, iters),
                "verified": gen_ver,
                "note": "the matrices are read once per solve; an iteration is two products (LDS-read bound) and two hand-offs "
                        "between the two workgroups of a problem (0.44-0.56 us each, tools/hop_probe.hip): the bound is that "
                        "latency, not HBM and not the VALU; us_per_iteration_of_a_cluster includes the tile loads (upper bound)"}

            # standalone SpMV over FOUR distinct 308 MB matrices in rotation (1.23 GB: the 256 MiB Infinity Cache
            # cannot hold anything between two uses of the same line), each launch bracketed by its own event pair
            mats = [S, P, S.clone(), P.clone()]
            x = torch.randn_like(gamma)
            y = torch.empty_like(gamma)

            def time_spmv(launches=104, rounds=3):
                res = []
                for _ in range(rounds):
                    evs = new_events(launches)
                    torch.cuda.synchronize()
                    for k, (e0, e1) in enumerate(evs):
                        e0.record(stream)
                        solver.spmv(n, N, B, mats[k % 4], x, y)
                        e1.record(stream)
                    torch.cuda.synchronize()
                    res.append(median([e0.elapsed_time(e1) for e0, e1 in evs[4:]]))
                return median(res)

            sp_bytes = spmv_bytes_per_launch(n, N, B, 4)
            sp_ms = time_spmv()
            sp_gbps = sp_bytes / (sp_ms * 1e-3) / 1e9
            # the same kernel when consecutive launches alternate between only two matrices (616 MB): what the
            # Infinity Cache adds
            mats2 = mats
            mats = [S, P, S, P]
            sp2_ms = time_spmv()
            mats = mats2
            solver.set_symmetric(1)   # only [D|R] is read (caller's word: a device check would cost as much as the product)
            sps_ms = time_spmv()
            solver.set_symmetric(2)
            del mats, mats2
            out["spmv_GBps"] = sp_gbps
            out["spmv"] = {"bound": "hbm", "kernel": "spmv_kernel<float,14,2,4>", "achieved": sp_gbps,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": sp_gbps / HBM_PEAK_GBPS,
                           "frac_of_copy_ceiling": sp_gbps / HBM_COPY_CEIL_GBPS,
                           "frac_of_cold_read_ceiling": sp_gbps / HBM_COLD_READ_GBPS,
                           "traffic": pmc_traffic("spmv_kernel<float,14"),
                           "algorithmic_bytes_per_launch": sp_bytes, "kernel_ms": sp_ms, "statistic": "median of 100 launches x 3 rounds",
                           "rotation": "4 matrices x 308 MB = 1.23 GB (Infinity Cache cannot serve it)",
                           "two_matrix_rotation": {"kernel_ms": sp2_ms, "GBps": sp_bytes / (sp2_ms * 1e-3) / 1e9,
                                                   "note": "616 MB rotation, as in round 1: includes Infinity-Cache hits"}}
            out["spmv_symmetric"] = {"kernel": "spmv_sym_kernel<float,14,4> (gbdpcg_set_symmetric(1): reads [D|R] only)",
                                     "achieved": sp_bytes / (sps_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": sp_bytes / (sps_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                     "traffic": pmc_traffic("spmv_sym_kernel<float,14"), "kernel_ms": sps_ms,
                                     "bytes_streamed_per_launch": B * ((2 * N - 1) * n * n + 2 * n * N) * 4,
                                     "frac_streamed": B * ((2 * N - 1) * n * n + 2 * n * N) * 4 / (sps_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        except Exception as e:   # noqa: BLE001 -- the headline above is already measured: report, do not lose the line
            out["secondary_blocks_error"] = f"{type(e).__name__}: {e}"
            solver.set_symmetric(2)

    graph.close()
    del S, P, gamma, lam, r, p
    torch.cuda.empty_cache()

    # The blocks behind the headline must not cost the line: a failure in one of them is reported in its place.
    def guarded(key, fn, *a):
        try:
            out[key] = fn(*a)
        except Exception as e:   # noqa: BLE001 -- reported, not swallowed: the key carries the error
            out[key] = {"error": f"{type(e).__name__}: {e}"}

    if world == 1 and not args.no_configs:
        guarded("configs", bench_configs, solver, torch, binding, synth, dev, stream)
        guarded("mpc_step", bench_mpc_step, solver, torch, binding, synth, dev, stream)
    if world == 1 and not args.no_cpu_baseline:
        guarded("cpu_baseline", cpu_baseline, n, N, iters)
    if rank == 0:
        print(json.dumps(out), flush=True)

    solver.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def bench_configs(solver, torch, binding, synth, dev, stream):
    """The other single-GPU BASELINE configs, each with the bound it has: C2 (n=14, N=64, fp32, one problem), C4
    (n=36, N=256, fp64, one problem), C5's 8192-problem batch on one GPU.  hipGraph replay, median of per-replay
    HIP-event times; 'fixed25' = exit_tol 0 / 25 iterations, 'converged' = tol 1e-6."""
    res = {}

    def new_events(k):
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]

    for name, n, N, B, dt, reps in (("C2", 14, 64, 1, torch.float32, 100), ("C4", 36, 256, 1, torch.float64, 100),
                                    ("C5_on_one_gpu", 14, 128, 8192, torch.float32, 20)):
        es = 4 if dt == torch.float32 else 8
        g = synth.gen_torch_seeded(n, N, 0, B, dev, dt, seed=BASE_SEED)
        S, gamma = g["S"], g["gamma"]
        del g
        P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
        lam = torch.zeros_like(gamma)
        r, p = torch.empty_like(gamma), torch.empty_like(gamma)
        it = torch.zeros(B, dtype=torch.int32, device=dev)
        fl = torch.zeros(B, dtype=torch.uint8, device=dev)
        rec = {"stateSize": n, "knotPoints": N, "batch": B, "dtype": "f32" if es == 4 else "f64",
               "path": {binding.PATH_FUSED: "fused", binding.PATH_SPLIT: "split",
                        binding.PATH_PERSISTENT: "persistent"}.get(solver.choose_path(es, n, N, B), "auto")}
        times, vers = {}, {}
        for tag, tol, mi in (("fixed5", 0.0, 5), ("fixed25", 0.0, MAX_ITER), ("converged", 1e-6, MAX_ITER)):
            gr = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, tol, mi, it, fl)
            it.fill_(-1)
            for _ in range(5):
                lam.zero_()
                gr.launch(stream)
            evs = new_events(reps)
            torch.cuda.synchronize()
            for e0, e1 in evs:
                lam.zero_()
                e0.record(stream)
                gr.launch(stream)
                e1.record(stream)
            torch.cuda.synchronize()
            vers[tag] = verify_solve(solver, torch, n, N, B, S, gamma, lam, it, fl, "fixed" if tol == 0.0 else "converged", mi)
            times[tag] = (median([a.elapsed_time(b) for a, b in evs]) * 1e3, float(it.float().mean()))
            gr.close()
        (t5, i5), (t25, i25), (tc, ic) = times["fixed5"], times["fixed25"], times["converged"]
        iter_bytes = 2 * (3 * N - 2) * n * n * es          # S and Pinv once per iteration, per problem
        floor_us = iter_bytes * B / (HBM_PEAK_GBPS * 1e9) * 1e6
        us_iter = (t25 - t5) / (i25 - i5)                  # cost of one more REAL iteration (both runs have exit_tol = 0)
        rec.update({"us_per_solve_fixed25": t25, "us_per_solve_fixed5": t5, "us_per_solve_converged": tc,
                    "iters_converged": ic, "us_per_iteration": us_iter,
                    "us_per_iteration_fixed25_incl_launch": t25 / i25,
                    "problem_iters_per_sec": B * i25 / (t25 * 1e-6),
                    "hbm_floor_us_per_iteration": floor_us, "frac_of_hbm_floor": floor_us / us_iter,
                    "statistic": f"median of {reps} graph replays; us_per_iteration = (fixed25 - fixed5) / 20",
                    "verified": {k: vers[k] for k in ("fixed25", "converged")}})
        if rec["path"] == "persistent":
            rec["bound"] = ("cross-CU latency: one launch, block-rows register-resident on 128 CUs, two in-kernel all-gathers of "
                            "{partial inner product, boundary knots} per iteration; no matrix byte moves after the first touch, "
                            "so the section-8d HBM floor is reported for completeness only")
        elif B == 1:
            rec["bound"] = ("latency: one problem, matrices resident on one CU for the whole solve; the HBM floor is the "
                            "section-8d stream time and is reported for completeness")
        else:
            tf = pcg_flops_per_launch(n, N, B, MAX_ITER) / (t25 * 1e-6) / 1e12
            rec["bound"] = "valu (resident symmetric kernel, as the headline)"
            rec["tflops"] = tf
            rec["frac_of_fp32_peak"] = tf / FP32_VECTOR_PEAK_TFLOPS
        res[name] = rec
        del S, P, gamma, lam, r, p
        torch.cuda.empty_cache()
    return res


def bench_mpc_step(solver, torch, binding, synth, dev, stream, nu=7, reps=30):
    """SURVEY 8f-4, the steps either side of the solve, at the headline batch shape (1024 problems, stateSize 14, controlSize 7,
    knotPoints 128, fp32): KKT blocks -> S, gamma, G^-1 (gbdpcg_form_schur) -> stair Phi^-1 + PCG to 1e-6
    (gbdpcg_form_pinv_solve) -> primal step (gbdpcg_recover_primal).  Each stage event-timed on the launch stream, median of
    `reps`; the two new stages are one pass over their operands, so their bound is HBM: achieved = (every input once + every
    output once) / time against the 8 TB/s spec.  The working set of a stage (0.3-0.75 GB) exceeds the 256 MiB Infinity Cache."""
    nx, N, B = N_STATE, N_KNOTS, BATCH_PER_GPU
    G, C, g, c = synth.kkt_torch(nx, nu, N, B, dev, torch.float32, seed=BASE_SEED)
    S, gamma, Ginv = solver.form_schur(nx, nu, N, B, G, C, g, c, stream=stream)
    Pinv, lam, z = torch.empty_like(S), torch.zeros_like(gamma), torch.empty_like(g)
    r, p = torch.empty_like(gamma), torch.empty_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    fl = torch.zeros(B, dtype=torch.uint8, device=dev)
    gr = solver.graph_form_pinv_solve(nx, N, B, S, Pinv, gamma, lam, r, p, 1e-6, 100, it, fl)
    gr_all = solver.graph_kkt_step(nx, nu, N, B, G, C, g, c, S, gamma, Ginv, Pinv, lam, r, p, 1e-6, 100, it, fl, z)

    def timed(fn):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for e0, e1 in evs:
            e0.record(stream)
            fn()
            e1.record(stream)
        torch.cuda.synchronize()
        return median([a.elapsed_time(b) for a, b in evs]) * 1e3

    def form():
        solver.form_schur(nx, nu, N, B, G, C, g, c, S=S, gamma=gamma, Ginv=Ginv, stream=stream)

    def solve():
        lam.zero_()
        gr.launch(stream)

    def recover():
        solver.recover_primal(nx, nu, N, B, Ginv, C, g, lam, z=z, stream=stream)

    def chain():            # the whole step as one graph (gbdpcg_graph_create_kkt_step)
        lam.zero_()
        gr_all.launch(stream)

    t_form, t_solve, t_rec, t_all = timed(form), timed(solve), timed(recover), timed(chain)
    sym = bool(solver.check_symmetric(nx, N, B, S).all())
    ok = int(fl.sum()) == 0 and bool(torch.isfinite(z).all())
    ver = verify_solve(solver, torch, nx, N, B, S, gamma, lam, it, fl, "converged")   # the last replay of the whole step
    by_form = (2 * G.numel() + C.numel() + g.numel() + c.numel() + S.numel() + gamma.numel()) * 4
    by_rec = (Ginv.numel() + C.numel() + g.numel() + lam.numel() + z.numel()) * 4
    res = {"shape": {"stateSize": nx, "controlSize": nu, "knotPoints": N, "batch": B, "dtype": "f32"},
           "form_schur": {"kernel": "schur_form_quad2_kernel<14,7> (fp32: one input buffer, two waves per SIMD)", "bound": "hbm", "us": t_form,
                          "algorithmic_bytes": by_form, "achieved": by_form / t_form / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": by_form / t_form / 1e3 / HBM_PEAK_GBPS, "traffic": schur_traffic("schur_form_quad")},
           "form_pinv_solve": {"us": t_solve, "iters_mean": float(it.float().mean()), "tol": 1e-6,
                               "note": "stair Phi^-1 formed from S + PCG to tolerance, one graph (the formation's symmetry "
                                       "verdicts replace the solve's own test launch)"},
           "recover_primal": {"kernel": "schur_recover_quad_kernel<float,14,7>", "bound": "hbm", "us": t_rec,
                              "algorithmic_bytes": by_rec, "achieved": by_rec / t_rec / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": by_rec / t_rec / 1e3 / HBM_PEAK_GBPS, "traffic": schur_traffic("schur_recover_quad_kernel")},
           "us_per_step_of_1024_problems": t_all, "kkt_systems_per_sec": B / (t_all * 1e-6),
           "S_symmetric_in_storage": sym, "all_converged_and_finite": ok, "verified": ver,
           "statistic": f"median of {reps} event-timed repetitions per stage, and of the whole step replayed as one hipGraph "
                        "(gbdpcg_graph_create_kkt_step, lambda reset to 0 before each replay)"}
    gr.close()
    gr_all.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
    ap.add_argument("--no-converged", action="store_true", help="skip the run-to-tolerance block of the roofline object (profiling passes)")
    ap.add_argument("--dry-run", action="store_true",
                    help="host logic only (rendezvous, sharding, aggregation) with gloo and no GPU; reports no value")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        if args.gpus > 1:
            launch_ranks(args.gpus)      # parent: no GPU call before or after
            return
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(world_env)
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                     f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) "
                     f"or run `python bench.py --gpus {args.gpus}` without a torch.distributed environment")
    if args.dry_run:
        run_dry(args, world, rank)
    else:
        run_rank(args, world, rank, local_rank)


if __name__ == "__main__":
    main()
