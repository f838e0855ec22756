#!/usr/bin/env python3
"""Times the two steps either side of the solve (csrc/schur.hip) next to Pinv formation and the solve itself, on
synthetic KKT blocks of the BASELINE batch shape:   python gbd-pcg_amd/tools/schur_run.py [--nx 14 --nu 7 --N 128 --batch 1024]
Algorithmic bytes = every input once + every output once (form: G, C, g, c -> S, gamma, G^-1; recover: G^-1, C, g, lambda -> z).
Inputs are drawn on the device (SPD cost blocks M M' + I, dynamics I + noise); event-timed, median of --reps."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gbd_pcg_amd import binding, synth  # noqa: E402


def kkt_on_device(nx, nu, N, B, dtype, seed=0):
    return synth.kkt_torch(nx, nu, N, B, "cuda", dtype, seed)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=14)
    ap.add_argument("--nu", type=int, default=7)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--stamps", action="store_true", help="library built with -DGBDPCG_SCHUR_STAMPS: phase boundaries of one step")
    a = ap.parse_args()
    dtype = torch.float32 if a.dtype == "f32" else torch.float64
    es = 4 if a.dtype == "f32" else 8
    nx, nu, N, B = a.nx, a.nu, a.N, a.batch
    s = binding.Solver(0)
    G, C, g, c = kkt_on_device(nx, nu, N, B, dtype)
    S, gamma, Ginv = s.form_schur(nx, nu, N, B, G, C, g, c)
    Pinv = torch.empty_like(S)
    lam = torch.zeros_like(gamma)
    z = torch.empty_like(g)
    it, fl = s.form_pinv_solve(nx, N, B, S, Pinv, gamma, lam, tol=1e-6, max_iter=100)
    torch.cuda.synchronize()
    sym = bool(s.check_symmetric(nx, N, B, S).all())
    out = {"shape": f"nx{nx} nu{nu} N{N} x{B} {a.dtype}", "S_symmetric_in_storage": sym, "iters_mean": float(it.float().mean()),
           "max_iter_exits": int(fl.sum())}
    by_form = (G.numel() * 2 + C.numel() + g.numel() + c.numel() + S.numel() + gamma.numel()) * es
    by_rec = (Ginv.numel() + C.numel() + g.numel() + lam.numel() + z.numel()) * es
    t = timed(lambda: s.form_schur(nx, nu, N, B, G, C, g, c, S=S, gamma=gamma, Ginv=Ginv), a.reps)
    out["form_schur_us"] = round(t * 1e3, 1)
    out["form_schur_GBps"] = round(by_form / t / 1e6, 0)
    t = timed(lambda: s.form_pinv(nx, N, B, S, Pinv=Pinv), a.reps)
    out["form_pinv_us"] = round(t * 1e3, 1)

    def solve():
        lam.zero_()
        s.solve(nx, N, B, S, Pinv, gamma, lam, tol=1e-6, max_iter=100, iters=it, max_iter_exit=fl)
    t = timed(solve, a.reps)
    out["solve_converged_us"] = round(t * 1e3, 1)
    t = timed(lambda: s.recover_primal(nx, nu, N, B, Ginv, C, g, lam, z=z), a.reps)
    out["recover_primal_us"] = round(t * 1e3, 1)
    out["recover_primal_GBps"] = round(by_rec / t / 1e6, 0)
    if a.stamps:
        gam2 = torch.zeros(gamma.numel() + 24, dtype=gamma.dtype, device=gamma.device)
        s.form_schur(nx, nu, N, B, G, C, g, c, S=S, gamma=gam2, Ginv=Ginv)
        torch.cuda.synchronize()
        st = gam2[-24:].view(torch.int64).cpu().tolist()[:10]
        # fp32 (one input buffer): the requests for the next step go out at the END of a step, behind its stores, and the S rows are
        # stored there too; fp64 (two buffers): requests at the top, S rows of the previous step stored during the elimination
        names = ["wait for the requests", "(fp64: issue next requests)", "fix-ups + columns into registers", "elimination (fp64: + previous S stores)",
                 "carry", "G^-1 in place, A / B from LDS, W, V products", "T product", "D, gamma", "G^-1 stores (fp32: + S, gamma stores, next requests)"]
        out["stamps_shader_cycles"] = {names[i]: st[i + 1] - st[i] for i in range(9)}
        out["stamps_step_total"] = st[9] - st[0]
    print(json.dumps(out))
    s.close()


if __name__ == "__main__":
    main()
