"""Wall-clock latency of the reference's own entry point, the blocking device-pointer overload
(solvePCG<T>(state_size, knot_points, d_S, d_Pinv, ...), include/interface.cuh:92-144), as an MPC loop
would call it once per control step: gbdpcg_solve_blocking_* = launch + status copy-back + stream sync.
Measured on one MI355X (round 1): C2 37 us, n=14 N=128 159 us, C4 196 us (split path: 53 launches, host-bound: a
hipGraph replay of the same solve is 191 us wall of which ~90 us is hipGraphLaunch itself), n=2 N=3 29 us.
Run on the GPU box: python gbd-pcg_amd/tools/blocking_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

solver = binding.Solver(0)
PATHS = {1: 'fused', 2: 'split', 3: 'persistent'}
for (name, n, N, dt) in [("C2 n=14 N=64 fp32", 14, 64, torch.float32), ("n=14 N=128 fp32", 14, 128, torch.float32),
                         ("C4 n=36 N=256 fp64", 36, 256, torch.float64), ("C1 n=2 N=3 fp64", 2, 3, torch.float64)]:
    g = synth.gen_torch(n, N, 1, "cuda", dt, seed=1234)
    S, gamma = g["S"], g["gamma"]
    P = solver.form_pinv(n, N, 1, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    r, p = torch.empty_like(gamma), torch.empty_like(gamma)
    for _ in range(20):
        lam.zero_()
        it, fl = solver.solve_blocking(n, N, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)
    ts = []
    for _ in range(200):
        lam.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it, fl = solver.solve_blocking(n, N, S, P, gamma, lam, r, p, tol=1e-6, max_iter=25)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"{name}: solve_blocking wall  median {ts[len(ts)//2]:7.1f} us  p10 {ts[len(ts)//10]:7.1f} us  ({it} iterations, "
          f"path {PATHS.get(solver.choose_path(S.element_size(), n, N, 1))})", flush=True)

# comparison: explicit graph replay + device synchronize, wall clock (config 4)
n, N, dt = 36, 256, torch.float64
g = synth.gen_torch(n, N, 1, "cuda", dt, seed=1234)
S, gamma = g["S"], g["gamma"]
P = solver.form_pinv(n, N, 1, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma)
it = torch.zeros(1, dtype=torch.int32, device="cuda")
fl = torch.zeros(1, dtype=torch.uint8, device="cuda")
gr = solver.graph_solve(n, N, 1, S, P, gamma, lam, None, None, 1e-6, 25, it, fl)
for mode, name in ((0, "graph replay + sync"), (1, "eager solve + sync")):
    ts = []
    for _ in range(100):
        lam.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 0:
            gr.launch()
        else:
            solver.solve(n, N, 1, S, P, gamma, lam, None, None, tol=1e-6, max_iter=25, iters=it, max_iter_exit=fl)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"C4 {name}: wall median {ts[50]:.1f} us", flush=True)
