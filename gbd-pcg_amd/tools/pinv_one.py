"""A few stair formations at the headline batch shape (for rocprofv3 passes): python pinv_one.py [n] [reps] [sym]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gbd_pcg_amd import binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = binding.Solver(0)
g = synth.gen_torch(n, 128, 1024, "cuda", torch.float32, seed=1)
P = torch.empty_like(g["S"])
for _ in range(reps):
    s.form_pinv(n, 128, 1024, g["S"], binding.PINV_STAIR, P)
torch.cuda.synchronize()
