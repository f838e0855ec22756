"""A/B of the one-launch stair kernel at the headline batch shape: MFMA products + one LDS-DMA request per workgroup (shipped for
stateSize 16; GBDPCG_PINV_MFMA_FROM=12 takes it from 12 on) vs the VALU 2 x 2 tiles (GBDPCG_PINV_NO_MFMA=1); read once per process:
each arm runs in a child.  Event-timed, median of 30, per state size."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r"""
import sys, json, torch
sys.path.insert(0, sys.argv[1])
from gbd_pcg_amd import binding, synth
s = binding.Solver(0)
out = {}
for n in (14, 12, 16):
    N, B = 128, 1024
    g = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=1)
    P = torch.empty_like(g["S"])
    for _ in range(5): s.form_pinv(n, N, B, g["S"], binding.PINV_STAIR, P)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); s.form_pinv(n, N, B, g["S"], binding.PINV_STAIR, P); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[15] * 1e3
    err = float((P - g["Pinv"]).norm() / g["Pinv"].norm())
    sym = int(s.check_symmetric(n, N, B, P).min())
    by = 2 * g["S"].numel() * 4
    out[n] = {"us": t, "GBps": by / t / 1e3, "rel_diff_vs_host_stair": err, "exactly_symmetric": sym}
print(json.dumps(out))
"""
for arm, env in (("mfma", {"GBDPCG_PINV_MFMA_FROM": "12"}), ("valu", {"GBDPCG_PINV_NO_MFMA": "1"})) * 3:
    o = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
    print(arm, line[-1] if line else o.stderr[-2000:], flush=True)
