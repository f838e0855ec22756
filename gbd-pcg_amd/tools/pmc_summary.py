#!/usr/bin/env python3
"""Mean per dispatch of every counter in rocprofv3 --pmc counter_collection CSVs, per kernel.
    python gbd-pcg_amd/tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "gbdpcg"
    files = [root] if os.path.isfile(root) else glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if want not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = k.split("(")[0]
        for (d, c), v in per_dispatch.items():
            acc[(names[d], c)].append(v)
    for (k, c), v in sorted(acc.items()):
        print(f"{k[:70]:70s} {c:24s} n={len(v):<4d} mean={sum(v) / len(v):.4e} max={max(v):.4e}")


if __name__ == "__main__":
    main()
