#!/usr/bin/env python3
"""Config 4 (n = 36, N = 256, fp64, one problem) on the persistent path: us per iteration = (t25 - t5) / 20 and the converged solve,
graph replay, median of 100; against the oracle's iteration count.  Run with GBDPCG_LIB=<variant> for A/B builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gbd_pcg_amd import binding, synth
s = binding.Solver(0)
n, N, B = 36, 256, 1
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float64, seed=1234)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma)
it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
out = {}
for tag, tol, mi in (("t5", 0.0, 5), ("t25", 0.0, 25), ("conv", 1e-6, 25)):
    gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, tol, mi, it, fl)
    for _ in range(10):
        lam.zero_(); gr.launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    torch.cuda.synchronize()
    for a, b in evs:
        lam.zero_(); a.record(); gr.launch(); b.record()
    torch.cuda.synchronize()
    out[tag] = sorted(a.elapsed_time(b) for a, b in evs)[50] * 1e3
    out[tag + "_it"] = int(it[0])
    gr.close()
y = s.spmv(n, N, B, S, lam)
res = float((gamma - y).norm() / gamma.norm())
print("C4 %s: us/iter %.2f  fixed25 %.1f us  converged %.1f us (%d its)  residual %.2e" % (os.environ.get("GBDPCG_LIB", "shipped").split("_")[-1], (out["t25"] - out["t5"]) / 20, out["t25"], out["conv"], out["conv_it"], res), flush=True)
