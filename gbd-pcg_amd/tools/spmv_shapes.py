#!/usr/bin/env python3
"""Block-tridiagonal SpMV by shape (gbdpcg_spmv, general storage): median us and GB/s of the algorithmic bytes, rotating over enough
matrices that the Infinity Cache cannot serve them.   python gbd-pcg_amd/tools/spmv_shapes.py [n,N,B,dtype ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

shapes = [a for a in sys.argv[1:] if "," in a] or ["%d,128,1024,f32" % n for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 36)]
s = binding.Solver(0)
s.set_symmetric(0)
for sh in shapes:
    n, N, B, dt = sh.split(",")
    n, N, B = int(n), int(N), int(B)
    dtype = torch.float32 if dt == "f32" else torch.float64
    es = 4 if dt == "f32" else 8
    per = 3 * n * n * N * B * es
    copies = max(2, int(1.3e9 // per) + 1)
    Ss = [torch.randn(3 * n * n * N * B, dtype=dtype, device="cuda") for _ in range(copies)]
    x = torch.randn(n * N * B, dtype=dtype, device="cuda")
    for i in range(copies):
        s.spmv(n, N, B, Ss[i], x)
    reps = 30
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(evs):
        a.record()
        s.spmv(n, N, B, Ss[i % copies], x)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3
    by = ((3 * N - 2) * n * n + 2 * n * N) * es * B
    print("n=%d N=%d batch=%d %s spmv: %8.1f us  %6.0f GB/s" % (n, N, B, dt, t, by / t / 1e3), flush=True)
    del Ss, x
    torch.cuda.empty_cache()
