#!/usr/bin/env python3
"""Where one iteration of the cluster kernel (csrc/pcg_cluster.hip) spends its time: cycle stamps left by block 0 in
iteration 3 of its first problem, read from the diagnostic build of the library.

    make -C gbd-pcg_amd/csrc fvariant NAME=clstamps EXTRA=-DGBDPCG_CL_STAMPS UNITS="pcg_cluster api"
    GBDPCG_LIB=gbd-pcg_amd/csrc/variants/libgbdpcg_clstamps.so python gbd-pcg_amd/tools/cluster_stamps.py [N=128] [B=128]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402

n = 14
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
s = binding.Solver(0)
s.set_symmetric(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32, seed=1234)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma)
fn = s.lib.gbdpcg_internal_cluster_ws
fn.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
print("cycles from the top of iteration 3 (polling wave of block 0): S p done | stores issued + gather done | barrier | "
      "updates + barrier | Pinv r .. decision + barrier | p update + barrier ;  wave 0: product done, publish issued")
for rep in range(10):
    lam.zero_()
    s.solve(n, N, B, S, P, gamma, lam, tol=0.0, max_iter=8)
    torch.cuda.synchronize()
    host = (ctypes.c_uint64 * 32)()
    assert hip.hipMemcpy(host, ctypes.c_void_p(fn(s.h)), 256, 2) == 0
    r = list(host)
    if rep >= 2:
        print("   ", [int(r[i] - r[1]) for i in (2, 3, 4, 5, 6, 7)], "   wave 0:", [int(r[i] - r[8]) for i in (9, 10)],
              "   tile loads of wave 0:", int(r[13] - r[12]), "cycles")
        if B > 3 * 128:
            print("        fourth problem of block 0, wave 0, us from its start (real-time clock): tiles in | windows + barrier | r, hand-off | "
                  "p, hand-off | iterations done | outputs + barrier:", [round((r[i] - r[14]) / 100.0, 2) for i in (15, 16, 17, 18, 19, 20)],
                  "  all problems of block 0: %.1f us" % ((r[22] - r[21]) / 100.0))

# every workgroup's start and end on the real-time clock (last solve)
wg = (ctypes.c_uint64 * 512)()
# kClCtrlBytes + two parities of 640-byte slots (pcg_cluster.hip)
assert hip.hipMemcpy(wg, ctypes.c_void_p(fn(s.h) + (256 + 256 * 16) + 2 * 512 * 640), 4096, 2) == 0
nb = min(256, 2 * min(B, 128))
st = [wg[2 * i] for i in range(nb)]
en = [wg[2 * i + 1] for i in range(nb)]
t0 = min(st)
dur = sorted((e - b) / 100.0 for b, e in zip(st, en))
print("workgroups: starts spread over %.1f us; end - start min / median / max = %.1f / %.1f / %.1f us; last end - first start = %.1f us"
      % ((max(st) - t0) / 100.0, dur[0], dur[len(dur) // 2], dur[-1], (max(en) - t0) / 100.0))
