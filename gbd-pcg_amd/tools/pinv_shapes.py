#!/usr/bin/env python3
"""Stair Phi^-1 formation by shape (gbdpcg_form_pinv): median us and GB/s of S read once + Pinv written once.
    python gbd-pcg_amd/tools/pinv_shapes.py [n,N,B,dtype ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

shapes = [a for a in sys.argv[1:] if "," in a] or ["%d,128,1024,f32" % n for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 36)]
s = binding.Solver(0)
for sh in shapes:
    n, N, B, dt = sh.split(",")
    n, N, B = int(n), int(N), int(B)
    dtype = torch.float32 if dt == "f32" else torch.float64
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", dtype, seed=1234)
    S = g["S"]
    del g
    P = torch.empty_like(S)
    for _ in range(3):
        s.form_pinv(n, N, B, S, binding.PINV_STAIR, P)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        s.form_pinv(n, N, B, S, binding.PINV_STAIR, P)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[7] * 1e3
    by = 2 * 3 * n * n * N * B * S.element_size()
    print("n=%d N=%d batch=%d %s stair: %8.1f us  %6.0f GB/s" % (n, N, B, dt, t, by / t / 1e3), flush=True)
    del S, P
    torch.cuda.empty_cache()
