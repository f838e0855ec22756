#!/usr/bin/env python3
"""Time per PCG iteration of the batched solve: fixed-iteration solves (exit_tol = 0) at several max_iter,
least-squares fit  t = t0 + iters * t_iter.   python gbd-pcg_amd/tools/iter_fit.py [n N batch] [--mode 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[14, 128, 1024])
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 fused, 2 split")
    args = ap.parse_args()
    n, N, B = args.shape
    solver = binding.Solver(0)
    g = synth.gen_torch(n, N, B, "cuda", torch.float32, seed=1234)
    S, gamma = g["S"], g["gamma"]
    P = solver.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    solver.set_symmetric(args.mode)
    solver.set_path(args.path)
    xs, ys = [], []
    for iters in (5, 10, 20, 40, 80):
        graph = solver.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0, iters, it, fl)
        ts = []
        for r in range(args.reps + 5):
            lam.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graph.launch(); b.record(); torch.cuda.synchronize()
            if r >= 5:
                ts.append(a.elapsed_time(b))
        graph.close()
        t = float(np.median(ts))
        xs.append(iters); ys.append(t)
        print(f"max_iter {iters:3d}: {t * 1e3:8.1f} us per batch solve")
    k, t0 = np.polyfit(xs, ys, 1)
    rounds = -(-B // 256)
    print(f"fit: t0 = {t0 * 1e3:.1f} us, per iteration {k * 1e3:.2f} us per batch "
          f"({k * 1e3 / rounds:.3f} us per problem-iteration on one CU at {rounds} rounds)")
    solver.set_symmetric(2)
    solver.close()


if __name__ == "__main__":
    main()
