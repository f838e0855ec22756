"""A few general-storage solves on the cluster path for a profiler to look at (rocprofv3 --kernel-trace --stats -- python this).
   python gbd-pcg_amd/tools/cluster_run.py [N=128] [B=1024] [max_iter=25]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/gbd-pcg_amd/", 1)[0])
from gbd_pcg_amd import binding, synth  # noqa: E402

n = 14
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 25
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma)
s.set_symmetric(0)
for _ in range(20):
    lam.zero_()
    s.solve(n, N, B, S, P, gamma, lam, tol=0.0, max_iter=mi)
torch.cuda.synchronize()
print("done")
