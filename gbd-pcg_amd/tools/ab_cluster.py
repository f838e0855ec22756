#!/usr/bin/env python3
"""A/B of the cluster kernel (general storage, gbdpcg_set_symmetric(0)): median graph-replay time at fixed iteration counts
-> fixed cost per batch and time per iteration, for each library given, interleaved on ONE device.
    python gbd-pcg_amd/tools/ab_cluster.py [N=128] [B=1024] base v1 v2 ...   (csrc/variants/libgbdpcg_<name>.so; base = shipped)
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
n, N, B = 14, int(sys.argv[2]), int(sys.argv[3])
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma); it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
s.set_symmetric(0)
out = {}
for iters, tol in ((5, 0.0), (25, 0.0), (25, 1e-6)):
    gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, tol, iters, it, fl)
    for _ in range(10): lam.zero_(); gr.launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    torch.cuda.synchronize()
    for a, b in evs: lam.zero_(); a.record(); gr.launch(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    out["conv" if tol else "fixed%d" % iters] = t[30]
    out["iters"] = float(it.float().mean())
    gr.close()
print(json.dumps(out))
"""


def main():
    args = sys.argv[1:]
    N = int(args.pop(0)) if args and args[0].isdigit() else 128
    B = int(args.pop(0)) if args and args[0].isdigit() else 1024
    names = args or ["base"]
    res = {k: [] for k in names}
    for rnd in range(2):
        for name in names:
            env = dict(os.environ)
            env.pop("GBDPCG_LIB", None)
            if name != "base":
                env["GBDPCG_LIB"] = os.path.join(ROOT, "gbd-pcg_amd", "csrc", "variants", f"libgbdpcg_{name}.so")
            out = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(N), str(B)], env=env, capture_output=True, text=True, timeout=300)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:])
                continue
            res[name].append(json.loads(line[-1]))
    rounds = -(-B // max(1, min(B, 256 // max(1, -(-N // 72)))))
    for name in names:
        for r in res[name]:
            per_it = (r["fixed25"] - r["fixed5"]) / 20 * 1e3
            fixed = r["fixed5"] * 1e3 - 5 * per_it
            print(f"{name:10s} N={N} B={B}: fixed5 {r['fixed5']*1e3:7.1f} us  fixed25 {r['fixed25']*1e3:7.1f} us  converged {r['conv']*1e3:7.1f} us ({r['iters']:.1f} it)"
                  f"  => {per_it:6.2f} us per iteration of the batch ({per_it / rounds:5.2f} us per round), {fixed:6.1f} us fixed")


if __name__ == "__main__":
    main()
