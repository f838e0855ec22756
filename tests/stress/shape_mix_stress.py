#!/usr/bin/env python3
"""Random shapes on ONE handle, back to back: every on-chip kernel family (single-workgroup resident, clusters of one to eight
members (fp64: four), persistent, persistent in slices) and the streaming ones take turns, in random symmetric modes, and
every problem is compared with the CPU oracle (the checker; this file lives under tests/ for that reason).  The cluster kernels
share one hand-off workspace across shapes: a stale granule of one shape must never satisfy a poll of another.
     python tests/stress/shape_mix_stress.py [cases=150] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
s = binding.Solver(0)
bad = 0
seen = {}
for c in range(cases):
    f64 = bool(rng.integers(0, 2))
    n = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24, 36]))
    if n >= 20:
        N, B = int(rng.integers(8, 140)), int(rng.choice([1, 2, 3, 5, 8]))
    else:
        per_wg = 8 * (64 // (n if (f64 or n % 2) else n // 2))
        N = int(rng.integers(2, min((4 if f64 else 8) * per_wg, 600) + 1))
        B = int(rng.choice([1, 2, 7, 33, 130, 300]))
    dt = np.float64 if f64 else np.float32
    es = 8 if f64 else 4
    mode = int(rng.choice([0, 1, 2]))
    d = synth.gen_numpy(n, N, seed=9000 + c, batch=min(B, 6), dtype=dt)
    idx = np.arange(B) % min(B, 6)
    S, P = d["S"][idx], d["Pinv"][idx]
    g = (d["gamma"][idx] * (1.0 + 0.01 * (np.arange(B) // 6))[:, None]).astype(dt)
    ob = orc.pcg_batch(n, N, B, S, P, g, tol=1e-6, max_iter=60, nthreads=8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    lam = torch.zeros(B, n * N, dtype=torch.float64 if f64 else torch.float32, device="cuda")
    s.set_symmetric(mode)
    it, fl = s.solve(n, N, B, t(S), t(P), t(g), lam, tol=1e-6, max_iter=60)
    torch.cuda.synchronize()
    s.set_symmetric(2)
    key = (s.choose_path(es, n, N, B), s.cluster_members(es, n, N))
    seen[key] = seen.get(key, 0) + 1
    di = int(np.abs(it.cpu().numpy().astype(np.int64) - ob["iters"].astype(np.int64)).max())
    lam = lam.cpu().numpy()
    err = max(np.linalg.norm(lam[b] - ob["lambda_"][b]) / np.linalg.norm(ob["lambda_"][b]) for b in range(B))
    ok = di <= (0 if f64 else 1) and err < (1e-10 if f64 else 2e-6) and not fl.cpu().numpy().any()
    if not ok:
        bad += 1
        print("BAD case", c, "n", n, "N", N, "B", B, dt.__name__, "mode", mode, "path/members", key, "iters diff", di, "err %.2e" % err, flush=True)
print("%d cases, %d bad; (path, cluster members) -> count: %s" % (cases, bad, dict(sorted(seen.items()))))
sys.exit(1 if bad else 0)
