#!/usr/bin/env python3
"""Alternate launches that put the cluster kernel at different places of the kernel sequence (a launch with
gbdpcg_set_symmetric(0): the cluster kernel alone; then a mixed batch in the default mode: check kernel, resident symmetric
kernel, cluster kernel for the last quarter) and compare every general-storage problem of the mixed launches with the CPU
oracle (the checker).  On one box the sequence once produced answers built from stale hand-off granules; tags carry the
launch number since.  Lives under tests/ because it uses the oracle (test infrastructure) as the checker.
     python tests/stress/cluster_mixed_stress.py [repeats=40]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/stress/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gbd_pcg_amd import binding, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import test_gpu_cluster as tc  # noqa: E402

orc.build()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s = binding.Solver(0)
n, N, mi = 14, 128, 3
d = synth.gen_numpy(n, N, seed=31, batch=8, dtype=np.float32)


def general_only(B):
    idx = np.arange(B) % 8
    tc.run(s, n, N, B, d["S"][idx], d["Pinv"][idx], d["gamma"][idx], tol=0.0, max_iter=mi, symmetric=0)


def mixed(B, general_from):
    idx = np.arange(B) % 8
    S, P, g = d["S"][idx].copy(), d["Pinv"][idx].copy(), d["gamma"][idx]
    S[general_from:] = tc.unsymmetrize(S[general_from:], n, N, B - general_from)
    out = tc.run(s, n, N, B, S, P, g, tol=0.0, max_iter=mi, symmetric=2)
    sub = np.arange(general_from, B)
    ob = orc.pcg_batch(n, N, len(sub), S[sub], P[sub], g[sub], tol=0.0, max_iter=mi)
    err = np.array([tc.relerr(out["lambda_"][b], ob["lambda_"][j]) for j, b in enumerate(sub)])
    return int((err > 1e-6).sum())


total = 0
for rep in range(reps):
    general_only(384 if rep % 2 == 0 else 256)
    bad = mixed(512, 384)
    total += bad
    if bad:
        print("   repeat", rep, ":", bad, "problems differ from the oracle")
print(f"{reps} x (general-only launch, mixed launch): {total} problems differ from the oracle")
sys.exit(1 if total else 0)
