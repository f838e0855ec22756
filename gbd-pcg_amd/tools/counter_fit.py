#!/usr/bin/env python3
"""Per-ITERATION share of SQ counters of the resident symmetric kernel: run it at max_iter 5 and 45 under
rocprofv3 --pmc and difference the per-dispatch counters (the load phase and the prologue cancel).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS \
        SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d <out> -- \
        python3 <repo>/gbd-pcg_amd/tools/counter_fit.py run
    python3 gbd-pcg_amd/tools/counter_fit.py digest <out>
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import torch
    from gbd_pcg_amd import binding, synth
    n, N, B = 14, 128, 1024
    s = binding.Solver(0)
    g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
    S, gamma = g["S"], g["gamma"]
    P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
    lam = torch.zeros_like(gamma)
    s.set_symmetric(1)
    for iters in (5, 45, 5, 45, 5, 45):
        lam.zero_()
        s.solve(n, N, B, S, P, gamma, lam, tol=0.0, max_iter=iters)
        torch.cuda.synchronize()


def digest(root):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pcg_resident_sym_kernel" in row["Kernel_Name"]:
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)
    short, long_ = [per[i] for i in ids[0::2]], [per[i] for i in ids[1::2]]
    names = sorted(short[0])
    print(f"{'counter':24s} {'max_iter 5':>14s} {'max_iter 45':>14s} {'per iteration':>14s} {'outside iterations':>18s}")
    res = {}
    for c in names:
        a = sum(d[c] for d in short) / len(short)
        b = sum(d[c] for d in long_) / len(long_)
        k = (b - a) / 40.0
        res[c] = (k, a - 5 * k)
        print(f"{c:24s} {a:14.4e} {b:14.4e} {k:14.4e} {a - 5 * k:18.4e}")
    if "SQ_LDS_BANK_CONFLICT" in res and "SQ_LDS_IDX_ACTIVE" in res:
        print("LDS bank-conflict share of LDS-array cycles: inside an iteration %.3f, outside (loads, prologue, write-back) %.3f"
              % (res["SQ_LDS_BANK_CONFLICT"][0] / res["SQ_LDS_IDX_ACTIVE"][0], res["SQ_LDS_BANK_CONFLICT"][1] / res["SQ_LDS_IDX_ACTIVE"][1]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        digest(sys.argv[2])
