"""Fits split-path solve time against the iteration count (per-iteration cost = 2 launches) for configs 4 and 2.
Run on the GPU box from the repo root: python gbd-pcg_amd/tools/split_iter_fit.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gbd_pcg_amd import binding, synth
solver = binding.Solver(0)
for (n, N, B, dt) in [(36, 256, 1, torch.float64), (14, 64, 1, torch.float32)]:
    g = synth.gen_torch(n, N, B, "cuda", dt, seed=1234)
    S, P, gamma = g["S"], g["Pinv"], g["gamma"]
    lam = torch.zeros_like(gamma); r = torch.empty_like(gamma); p = torch.empty_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    solver.set_path(binding.PATH_SPLIT)
    for iters in (1, 5, 10, 15, 20, 25, 40):
        graph = solver.graph_solve(n, N, B, S, P, gamma, lam, r, p, 0.0, iters, it, fl)
        ts = []
        for rep in range(30):
            lam.zero_(); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); graph.launch(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"n={n} N={N} split fixed iters={iters:3d}: median {ts[len(ts)//2]:8.1f} us  min {ts[0]:8.1f} us  finite={bool(torch.isfinite(lam).all())} |lam|={lam.norm().item():.3e}", flush=True)
        graph.close()
