#!/usr/bin/env python3
"""Where a ROUND of the resident symmetric kernel (csrc/pcg_resident_sym.hip) spends its time: real-time-clock stamps left by
wave 0 of every workgroup at the phase boundaries of each of its problems, read from the diagnostic build of the library.

    make -C gbd-pcg_amd/csrc fvariant NAME=rsstamps EXTRA=-DGBDPCG_RS_STAMPS UNITS="pcg_resident_sym"
    GBDPCG_LIB=gbd-pcg_amd/csrc/variants/libgbdpcg_rsstamps.so python gbd-pcg_amd/tools/rs_stamps.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

n, N, B = 14, 128, 1024
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32, seed=1234)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma)
it = torch.zeros(B, dtype=torch.int32, device="cuda")
fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
fn = s.lib.gbdpcg_internal_cluster_ws
fn.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
s.set_symmetric(1)
names = ["loads issued", "tiles parked + picked", "vectors in, barrier", "prologue (2 products)", "iterations", "write-back + barrier"]
for label, tol, mi in (("25 fixed iterations", 0.0, 25), ("to 1e-6 (9 iterations)", 1e-6, 25)):
    gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, tol, mi, it, fl)
    for _ in range(6):
        lam.zero_()
        gr.launch()
    torch.cuda.synchronize()
    host = (ctypes.c_uint64 * (256 * 8 * 8))()
    assert hip.hipMemcpy(host, ctypes.c_void_p(fn(s.h) + 8192), 256 * 8 * 8 * 8, 2) == 0
    st = np.array(list(host), dtype=np.float64).reshape(256, 8, 8) / 100.0      # us
    t0 = st[:, 0, 0].min()
    print(f"== {label}: mean iterations {float(it.float().mean()):.1f}; kernel span (first start to last end) {st[:, 3, 6].max() - t0:.1f} us")
    for rnd in range(4):
        d = np.diff(st[:, rnd, :7], axis=1)
        med = np.median(d, axis=0)
        print(f"   round {rnd}: start {np.median(st[:, rnd, 0]) - t0:7.1f} us after the kernel's; " +
              " | ".join(f"{nm} {v:.1f}" for nm, v in zip(names, med)) + f" | whole round {np.median(st[:, rnd, 6] - st[:, rnd, 0]):.1f} us"
              f" (slowest workgroup {np.max(st[:, rnd, 6] - st[:, rnd, 0]):.1f})")
    gr.close()
