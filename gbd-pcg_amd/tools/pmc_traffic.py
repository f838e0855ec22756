#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from rocprofv3 PMC passes (counter_collection CSVs):

    pmc_traffic.py <bench FETCH_SIZE dir> <bench WRITE_SIZE dir> <bw_probe FETCH_SIZE dir> <out.json>

FETCH_SIZE is reported in KiB and, on gfx950, at about half the bytes of a wide streaming read
(MI355X_MICROARCH.md, HBM section); the factor is re-calibrated here on bw_probe's pure-read kernels,
whose byte count is known, in every access shape / cache policy the library uses.  WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

N_, KNOTS, BATCH, ITERS = 14, 128, 1024, 25
MAT = 3 * N_ * N_ * KNOTS * 4 * BATCH               # one matrix of the batch, full storage
ALG_SPMV = ((3 * KNOTS - 2) * N_ * N_ + 2 * N_ * KNOTS) * 4 * BATCH
ALG_PCG = BATCH * ((2 * ITERS + 2) * (3 * KNOTS - 2) * N_ * N_ + 5 * N_ * KNOTS) * 4
SYM_ONCE = BATCH * (2 * (2 * KNOTS - 1) * N_ * N_ + 5 * N_ * KNOTS) * 4   # [D|R] of S and Pinv read once per solve
FULL_ONCE = BATCH * (2 * 3 * KNOTS * N_ * N_ + 5 * N_ * KNOTS) * 4         # [L|D|R] of S and Pinv read once per solve


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            per[row["Dispatch_Id"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for d, v in per.items():
            acc[names[d]].append(v)
    return acc


def pick(acc, sub):
    for k, v in acc.items():
        if sub in k:
            # fixed-iteration solves only: drop the short dispatches (converged / no-op launches)
            big = [x for x in v if x > 0.5 * max(v)] or v
            return sum(big) / len(big), len(big)
    return None, 0


def main():
    fdir, wdir, cdir, out = sys.argv[1:5]
    fetch, write, cal = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE"), per_kernel(cdir, "FETCH_SIZE")
    res = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE, on "
                   "`bench.py --steps 3 --warmup 1` (config 3) and on gbd-pcg_amd/tools/bw_probe; MI355X. Counter unit = "
                   "KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide "
                   "streaming read; calibrated below on known-size pure reads in the access shapes and cache policies "
                   "the library uses; read bytes = FETCH_SIZE * 1024 / ratio(0.5). WRITE_SIZE is exact. FETCH_SIZE "
                   "counts Infinity-Cache hits too: it is the traffic leaving L2, not HBM traffic.",
           "calibration": {}, "kernels": {}}
    for k, v in cal.items():
        if "read" in k and "kernel" in k:
            name = k.split("(")[0].replace("void ", "")
            res["calibration"][name] = {"known_bytes": MAT, "FETCH_SIZE_KiB": sum(v) / len(v),
                                        "ratio": sum(v) / len(v) * 1024 / MAT, "dispatches": len(v)}
    table = [("pcg_resident_sym_kernel", "pcg_resident_sym_kernel<14,true> (config 3, 25 iterations, matrices resident on the CU)", ALG_PCG, SYM_ONCE),
             ("pcg_cluster_kernel<14", "pcg_cluster_kernel<14,2,true> (config 3, 25 iterations, general storage resident over two CUs per problem)", ALG_PCG, FULL_ONCE),
             ("pcg_fused_kernel<float, 14, 2, 8, false>", "pcg_fused_kernel<float,14,2,8,false> (config 3, 25 iterations, general)", ALG_PCG, ALG_PCG),
             ("check_symmetric_pair_kernel", "check_symmetric_pair_kernel<float> (config 3, S and Pinv)", None, 2 * BATCH * (KNOTS - 1) * 2 * N_ * N_ * 4),
             ("spmv_kernel<float, 14", "spmv_kernel<float,14,2,4> (config 3)", ALG_SPMV, ALG_SPMV),
             ("spmv_sym_kernel<float, 14", "spmv_sym_kernel<float,14,4> (config 3)", ALG_SPMV,
              BATCH * ((2 * KNOTS - 1) * N_ * N_ + 2 * N_ * KNOTS) * 4)]
    for sub, label, alg, need in table:
        f, nf = pick(fetch, sub)
        w, nw = pick(write, sub)
        if f is None or w is None:
            continue
        traffic = 2 * f * 1024 + w * 1024
        rec = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": traffic, "dispatches": nf,
               "bytes_the_kernel_must_move": need, "traffic_over_must_move": traffic / need}
        if alg:
            rec["algorithmic_bytes_per_launch"] = alg
            rec["traffic_over_algorithmic"] = traffic / alg
        res["kernels"][label] = rec
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
