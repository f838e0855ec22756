"""A few stair formations of the config-3 batch for a profiler to look at (rocprofv3 ... -- python this)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/gbd-pcg_amd/", 1)[0])
from gbd_pcg_amd import binding, synth  # noqa: E402

solver = binding.Solver(0)
n, N, B = 14, 128, 1024
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
P = torch.empty_like(g["S"])
for _ in range(10):
    solver.form_pinv(n, N, B, g["S"], binding.PINV_STAIR, P)
torch.cuda.synchronize()
print("done")
