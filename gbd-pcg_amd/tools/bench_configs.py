#!/usr/bin/env python3
"""Secondary measurements: every single-GPU BASELINE.json config, both solve paths where they apply.

    python gbd-pcg_amd/tools/bench_configs.py [--reps 50]

For each config: fixed-iteration solve (exit_tol = 0, max_iter = 25) replayed from a hipGraph,
median of per-replay HIP-event times; converge-to-1e-6 solve (iterations, time); batched SpMV.
Prints one JSON line per measurement.  bench.py stays the headline (config 3) benchmark.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

CONFIGS = [
    ("C2", 14, 64, 1, torch.float32),
    ("C3", 14, 128, 1024, torch.float32),
    ("C4", 36, 256, 1, torch.float64),
    ("C2x64", 14, 64, 64, torch.float32),
    ("C4x16", 36, 256, 16, torch.float64),
    ("C5on1", 14, 128, 8192, torch.float32),   # config 5's whole batch on one GPU
]


def timed(fn, reps, warmup=5):
    for _ in range(warmup):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    solver = binding.Solver(0)
    for name, n, N, B, dt in CONFIGS:
        if args.only and name not in args.only.split(","):
            continue
        es = 4 if dt == torch.float32 else 8
        g = synth.gen_torch(n, N, B, "cuda", dt, seed=1234)
        S, gamma = g["S"], g["gamma"]
        P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)   # exactly symmetric storage (default path streams [D|R])
        del g
        lam = torch.zeros_like(gamma)
        r, p = torch.empty_like(gamma), torch.empty_like(gamma)
        it = torch.zeros(B, dtype=torch.int32, device="cuda")
        fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
        mat = (3 * N - 2) * n * n * es
        # "general": the fused path with gbdpcg_set_symmetric(0) -- what storage that is not bit-symmetric gets (the cluster
        # kernel where the shape has it, else the kernel that streams both matrices every iteration)
        for path, pname in ((binding.PATH_FUSED, "fused"), (binding.PATH_FUSED, "general"), (binding.PATH_SPLIT, "split"),
                            (binding.PATH_PERSISTENT, "persist"), (binding.PATH_PERSISTENT_1R, "persist1r")):
            solver.set_path(path)
            solver.set_symmetric(0 if pname == "general" else 2)
            chosen = solver.choose_path(es, n, N, B)
            if chosen != path:
                continue  # forced path does not fit this shape
            if pname == "general" and B == 1 and solver.cluster_members(es, n, N) == 0:
                continue  # single problems without a cluster form: the same kernel as "fused"
            for tol, iters, tag in ((0.0, 5, "fixed5"), (0.0, 25, "fixed25"), (1e-6, 25, "tol1e-6")):
                graph = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, tol, iters, it, fl)

                def step():
                    lam.zero_()
                    graph.launch()
                it.fill_(-1)  # a replay that solves nothing leaves these behind (it must not inherit an earlier run's counts)
                med, best = timed(step, args.reps)
                assert int(it.min()) >= 1 and bool(torch.isfinite(lam).all()), "the timed replays did not solve anything"
                done = it.float().mean().item()
                rec = dict(config=name, n=n, N=N, batch=B, dtype=str(dt).replace("torch.", ""), path=pname, run=tag,
                           ms_median=med, ms_best=best, iters_mean=done,
                           problem_iters_per_s=B * done / (med * 1e-3),
                           us_per_iteration=med * 1e3 / max(done, 1),
                           algorithmic_GBps=B * ((2 * done + 2) * mat + 5 * n * N * es) / (med * 1e-3) / 1e9)
                print(json.dumps(rec), flush=True)
                graph.close()
        solver.set_path(binding.PATH_AUTO)
        solver.set_symmetric(2)
        x = torch.randn_like(gamma)
        y = torch.empty_like(gamma)
        k = [0]

        def sp():
            solver.spmv(n, N, B, S if k[0] % 2 == 0 else P, x, y)
            k[0] += 1
        med, best = timed(sp, args.reps)
        print(json.dumps(dict(config=name, n=n, N=N, batch=B, run="spmv", ms_median=med, ms_best=best,
                              algorithmic_GBps=B * ((3 * N - 2) * n * n + 2 * n * N) * es / (med * 1e-3) / 1e9)),
              flush=True)
    solver.close()


if __name__ == "__main__":
    main()
