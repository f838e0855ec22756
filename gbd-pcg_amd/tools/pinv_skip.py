"""Timing builds of pinv_stair_mfma_kernel with parts switched off (make fvariant NAME=pskipK EXTRA=-DGBDPCG_PINV_SKIP=K UNITS=pinv;
1 no elimination, 2 no products, 4 no result stores; results are WRONG): where the time of the stair kernel goes."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from gbd_pcg_amd import binding, synth
s = binding.Solver(0)
n, N, B = 14, 128, 1024
g = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=1)
P = torch.empty_like(g["S"])
for _ in range(5): s.form_pinv(n, N, B, g["S"], binding.PINV_STAIR, P)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
torch.cuda.synchronize()
for a, b in evs:
    a.record(); s.form_pinv(n, N, B, g["S"], binding.PINV_STAIR, P); b.record()
torch.cuda.synchronize()
print("US %.1f" % (sorted(a.elapsed_time(b) for a, b in evs)[15] * 1e3))
"""
for name in ("", "pskip1", "pskip2", "pskip4", "pskip3", "pskip7"):
    env = dict(os.environ)
    if name:
        env["GBDPCG_LIB"] = os.path.join(ROOT, "gbd-pcg_amd", "csrc", "variants", f"libgbdpcg_{name}.so")
    o = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=300)
    print(name or "shipped", [ln for ln in o.stdout.splitlines() if ln.startswith("US")] or o.stderr[-500:], flush=True)
