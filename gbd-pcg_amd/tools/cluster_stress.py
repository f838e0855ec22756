"""Repeat general-storage solves on the cluster path and count the launches whose iteration counts differ from the first
one (a hand-off race shows up as a different count, never as a hang: every spin is bounded).
   python gbd-pcg_amd/tools/cluster_stress.py [N] [B] [repeats]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/gbd-pcg_amd/", 1)[0])
from gbd_pcg_amd import binding, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n = 14
d = synth.gen_numpy(n, N, seed=500 + N, batch=B, dtype=np.float32)
s = binding.Solver(0)
s.set_symmetric(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dS, dP, dg = t(d["S"]), t(d["Pinv"]), t(d["gamma"])
ref = None
bad = 0
for i in range(reps):
    lam = torch.zeros_like(dg)
    it, fl = s.solve(n, N, B, dS, dP, dg, lam, None, None, tol=1e-6, max_iter=100)
    torch.cuda.synchronize()
    it = it.cpu().numpy().astype(np.int64)
    if ref is None:
        ref = it.copy()
        print("first launch:", it[:16], "flags", fl.cpu().numpy()[:16])
    elif not np.array_equal(it, ref):
        bad += 1
        if bad <= 5:
            print("launch", i, "differs:", it[:16], "flags", fl.cpu().numpy()[:16])
print(f"N={N} B={B}: {bad} of {reps - 1} repeats differ from the first launch")
