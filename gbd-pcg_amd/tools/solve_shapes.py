#!/usr/bin/env python3
"""Converged and fixed-count solve times by shape, default (AUTO) path, graph replay, median of --reps:
    python gbd-pcg_amd/tools/solve_shapes.py [n,N,B,dtype ...]      e.g.  12,128,1024,f32 14,128,1024,f64
Problems Gen(n, N, 1234 + i, 0.5), stair Phi^-1 formed on the device.  Prints one JSON record per shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

shapes = [a for a in sys.argv[1:] if "," in a] or ["12,128,1024,f32", "14,128,1024,f32", "14,128,1024,f64", "13,128,1024,f32",
                                                     "16,128,1024,f32", "18,128,1024,f32", "10,128,1024,f32", "8,256,1024,f32"]
reps = 30
s = binding.Solver(0)
for sh in shapes:
    n, N, B, dt = sh.split(",")
    n, N, B = int(n), int(N), int(B)
    dtype = torch.float32 if dt == "f32" else torch.float64
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", dtype, seed=1234)
    S, gamma = g["S"], g["gamma"]
    del g
    P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    rec = {"n": n, "N": N, "batch": B, "dtype": dt, "path": s.choose_path(4 if dt == "f32" else 8, n, N, B),
           "cluster_members": s.cluster_members(4 if dt == "f32" else 8, n, N)}
    for tag, tol, mi in (("converged", 1e-6, 100), ("fixed25", 0.0, 25)):
        gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, tol, mi, it, fl)
        for _ in range(3):
            lam.zero_()
            gr.launch()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for a, b in evs:
            lam.zero_()
            a.record()
            gr.launch()
            b.record()
        torch.cuda.synchronize()
        rec[f"us_{tag}"] = round(sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3, 1)
        rec[f"iters_{tag}"] = round(float(it.float().mean()), 2)
        y = s.spmv(n, N, B, S, lam)
        rec[f"resid_{tag}"] = float(((gamma - y).double().reshape(B, -1).norm(dim=1) / gamma.double().reshape(B, -1).norm(dim=1)).max())
        gr.close()
    print(json.dumps(rec), flush=True)
    del S, P, gamma, lam
    torch.cuda.empty_cache()
