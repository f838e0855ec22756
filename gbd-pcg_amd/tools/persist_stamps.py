#!/usr/bin/env python3
"""Where one phase of the persistent kernel (csrc/pcg_persist.hip) spends its time: cycle stamps left by workgroup 1 in
iteration 3 of a config-4 solve, read from the diagnostic build of the library.

    make -C gbd-pcg_amd/csrc variant NAME=stamps EXTRA=-DGBDPCG_PERSIST_STAMPS
    GBDPCG_LIB=gbd-pcg_amd/csrc/variants/libgbdpcg_stamps.so [GBDPCG_PERSIST_K=2] python gbd-pcg_amd/tools/persist_stamps.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gbd_pcg_amd import binding, synth  # noqa: E402


def main():
    n, N = 36, 256
    s = binding.Solver(0)
    one_r = len(sys.argv) > 1 and sys.argv[1] == "1r"
    s.set_path(binding.PATH_PERSISTENT_1R if one_r else binding.PATH_PERSISTENT)
    g = synth.gen_torch_seeded(n, N, 0, 1, "cuda", torch.float64, seed=1234)
    S, gamma = g["S"], g["gamma"]
    P = s.form_pinv(n, N, 1, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    rows = []
    for rep in range(12):
        lam.zero_()
        s.solve(n, N, 1, S, P, gamma, lam, tol=0.0, max_iter=8)
        torch.cuda.synchronize()
        fn = s.lib.gbdpcg_internal_persist_ws
        fn.restype = ctypes.c_void_p
        ptr = fn(s.h)
        host = (ctypes.c_uint64 * 16)()
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(host, ctypes.c_void_p(ptr), 128, 2) == 0
        rows.append(list(host))
    if one_r:
        print("single-reduction form, iteration 3, wave 0 of workgroup 1: cycles from the top of the iteration to "
              "[u done + barrier, w done + barrier, all-gather done, decision + barrier, updates + barrier]")
        for r in rows[2:]:
            print("   ", [int(r[i] - r[2]) for i in (3, 4, 5, 6, 7)])
        return
    # hop latency: every workgroup's real-time clock (100 MHz, common to the chip) at its publish and at the end of its sweep
    K = int(os.environ.get("GBDPCG_PERSIST_K", "2"))
    Wn = (N + K - 1) // K
    words = 32 + 2 * N * 16 + 2 * N * 2 * n * 2          # ctrl (kPersistCtrl) + partial slots (128-byte stride) + first halo region (fp64), in u64
    xs = (ctypes.c_uint64 * (2 * Wn))()
    assert hip.hipMemcpy(xs, ctypes.c_void_p(ptr + 8 * words), 8 * 2 * Wn, 2) == 0
    pub = [xs[2 * i] for i in range(Wn)]
    det = [xs[2 * i + 1] for i in range(Wn)]
    missing = [i for i in range(Wn) if pub[i] == 0 or det[i] == 0]
    if missing:   # a workgroup that left no stamp (not expected): say so instead of printing differences against zero
        print("no stamp from workgroups", missing[:8], "..." if len(missing) > 8 else "", "(%d of %d)" % (len(missing), Wn))
        keep = [i for i in range(Wn) if i not in set(missing)]
        pub, det = [pub[i] for i in keep], [det[i] for i in keep]
        Wn = len(keep)
    last = max(pub)
    print("hop (iteration 3, S p phase, last solve): publishes spread over %d x 10 ns; sweep ends %d .. %d x 10 ns after the LAST publish "
          "(median %d)" % (last - min(pub), min(det) - last, max(det) - last, sorted(det)[Wn // 2] - last))
    for tag, b in (("direction (S p)", 2), ("precond (Pinv r)", 8)):
        print(tag, ": cycles from phase start to [product+barrier, reduce+publish, sweep done, barrier]")
        for r in rows[2:]:
            t = r[b:b + 5]
            print("   ", [int(x - t[0]) for x in t[1:]])
    print("direction phase, wave 0: cycles from phase start to [row dot done, owner writes + wave sum done, at barrier, past barrier]")
    for r in rows[2:]:
        print("   ", [int(r[i] - r[2]) for i in (1, 7, 13, 3)])
    # shader clock vs 100 MHz real-time clock between the two phase starts
    for r in rows[2:5]:
        print("  phase-to-phase: %d cycles, %d x 10 ns" % (r[8] - r[2], r[15] - r[14]))


if __name__ == "__main__":
    main()
