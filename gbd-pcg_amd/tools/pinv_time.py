import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gbd_pcg_amd import binding, synth
solver = binding.Solver(0)
for (n, N, B, dt) in [(14, 128, 1024, torch.float32), (14, 64, 1, torch.float32), (36, 256, 1, torch.float64), (36, 256, 16, torch.float64), (36, 64, 256, torch.float32), (14, 128, 1024, torch.float64)]:
    g = synth.gen_torch(n, N, B, "cuda", dt, seed=1)
    P = torch.empty_like(g["S"])
    for kind, nm in ((binding.PINV_BLOCK_JACOBI, "jacobi"), (binding.PINV_STAIR, "stair")):
        for _ in range(3): solver.form_pinv(n, N, B, g["S"], kind, P)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record(); solver.form_pinv(n, N, B, g["S"], kind, P); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)[5]
        err = (P - g["Pinv"]).norm() / g["Pinv"].norm() if kind == binding.PINV_STAIR else 0.0
        print(f"n={n} N={N} batch={B} {dt} {nm}: {t*1e3:.1f} us   rel.diff vs torch stair {float(err):.2e}", flush=True)
