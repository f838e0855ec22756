#!/usr/bin/env python3
"""A/B of the resident symmetric kernel (config 3, 25 fixed iterations, caller-asserted symmetry: the kernel alone):
median per-replay time for each library given, interleaved over rounds on ONE device.
    python gbd-pcg_amd/tools/ab_resident.py base st1 st2 ...      (names of csrc/variants/libgbdpcg_<name>.so; base = shipped)
Each library runs in its own child process (the binding loads one library per process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, json, torch
sys.path.insert(0, sys.argv[1])
from gbd_pcg_amd import binding, synth
n, N, B, iters = 14, 128, 1024, int(sys.argv[2])
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma); it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
s.set_symmetric(1)
gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0 if iters == 25 else 1e-6, 25, it, fl)
for _ in range(10): lam.zero_(); gr.launch()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
torch.cuda.synchronize()
for a, b in evs: lam.zero_(); a.record(); gr.launch(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in evs)
print(json.dumps({"median_ms": t[50], "min_ms": t[0], "iters": float(it.float().mean())}))
"""


def main():
    names = sys.argv[1:] or ["base"]
    res = {k: [] for k in names}
    for rnd in range(3):
        for name in names:
            env = dict(os.environ)
            env.pop("GBDPCG_LIB", None)
            if name != "base":
                env["GBDPCG_LIB"] = os.path.join(ROOT, "gbd-pcg_amd", "csrc", "variants", f"libgbdpcg_{name}.so")
            out = subprocess.run([sys.executable, "-c", CHILD, ROOT, "25"], env=env, capture_output=True, text=True, timeout=300)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:])
                continue
            res[name].append(json.loads(line[-1])["median_ms"])
    for name in names:
        v = sorted(res[name])
        print(f"{name:10s} median-of-rounds {v[len(v) // 2] * 1e3:7.1f} us   rounds {[round(x * 1e3, 1) for x in res[name]]}")


if __name__ == "__main__":
    main()
