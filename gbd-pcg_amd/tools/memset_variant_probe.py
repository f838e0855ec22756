#!/usr/bin/env python3
"""Does the round-1 form of the flag / done-word initialisation (hipGraph MEMSET nodes instead of fill kernels) still
misbehave?  Run against the diagnostic variant and against the shipped library:

    make -C gbd-pcg_amd/csrc fvariant NAME=memset EXTRA=-DGBDPCG_FILL_WITH_MEMSET UNITS=symcheck
    GBDPCG_LIB=gbd-pcg_amd/csrc/variants/libgbdpcg_memset.so python gbd-pcg_amd/tools/memset_variant_probe.py
    python gbd-pcg_amd/tools/memset_variant_probe.py

Two graphs that contain such a node are replayed back to back with tensor.zero_() (an eager memset on the same
stream) in between, and what every single replay produced is recorded on the device:
  A  split path, n=36 N=256 fp64, one problem: the per-problem `done` word is cleared by the node; a replay that
     starts with a non-zero `done` word is a string of no-op launches and leaves iters / lambda untouched
  B  two-launch symmetry check (matrices only 8-byte aligned, so the one-launch pair kernel is not used): the flags are
     set to 1 by the node, then cleared where a pair mismatches; garbage flags send problems to the general kernel
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402


def main():
    s = binding.Solver(0)
    print("library:", binding.LIB_PATH)
    # ---- A: split path done words
    n, N = 36, 256
    g = synth.gen_torch_seeded(n, N, 0, 1, "cuda", torch.float64)
    S, gamma = g["S"], g["gamma"]
    P = s.form_pinv(n, N, 1, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    fl = torch.zeros(1, dtype=torch.uint8, device="cuda")
    s.set_path(binding.PATH_SPLIT)
    gr = s.graph_solve(n, N, 1, S, P, gamma, lam, None, None, 1e-6, 25, it, fl)
    R = 60
    log_it = torch.full((R,), -7, dtype=torch.int32, device="cuda")
    log_nrm = torch.zeros(R, dtype=torch.float64, device="cuda")
    for r in range(R):
        lam.zero_()
        it.fill_(-1)
        gr.launch()
        log_it[r:r + 1].copy_(it)          # queued behind the replay on the same stream
        log_nrm[r:r + 1].copy_(lam.norm().reshape(1))
    torch.cuda.synchronize()
    its = log_it.cpu().numpy()
    nr = log_nrm.cpu().numpy()
    bad = int((its != its[0]).sum() + (its <= 0).sum() + (np.abs(nr - nr[0]) > 1e-9 * nr[0]).sum())
    print(f"A split-path done words : iterations per replay {sorted(set(its.tolist()))}, |lambda| spread "
          f"{(nr.max() - nr.min()) / nr.max():.1e} -> {bad} bad replays of {R}")
    gr.close()
    s.set_path(binding.PATH_AUTO)
    # ---- B: two-launch symmetry check
    n, N, B = 14, 128, 512
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)

    def shifted(t):                       # 8-byte aligned view: the pair kernel needs 16
        buf = torch.zeros(t.numel() + 2, dtype=t.dtype, device="cuda")
        v = buf[2:]
        v.copy_(t.reshape(-1))
        assert v.data_ptr() % 16 == 8
        return v
    S = shifted(g["S"])
    P0 = s.form_pinv(n, N, B, g["S"], binding.PINV_STAIR)
    P = shifted(P0)
    gamma = g["gamma"]
    lam = torch.zeros_like(gamma)
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
    gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0, 10, it, fl)

    def per_replay(sync, reps=30):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(reps):
            lam.zero_()
            gr.launch()
            if sync:
                torch.cuda.synchronize()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps
    per_replay(True, 5)
    synced, queued = per_replay(True), per_replay(False)
    print(f"B two-launch check flags : {synced * 1e3:.0f} us per replay synchronised, {queued * 1e3:.0f} us queued back to back "
          f"(a queued replay that is several times slower = problems fell to the general kernel)")
    gr.close()
    s.close()


if __name__ == "__main__":
    main()
