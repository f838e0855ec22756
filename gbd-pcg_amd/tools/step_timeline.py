#!/usr/bin/env python3
"""Where a default config-3 step goes, kernel by kernel: digest of a rocprofv3 --kernel-trace of bench.py.
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $REPO/bench.py --no-configs --no-cpu-baseline
   python3 gbd-pcg_amd/tools/step_timeline.py /tmp/tl
Prints, for the last 50 (check, resident, general) triples of the trace: median duration of each kernel and of the gaps
between them, and the median distance from one step's first kernel to the next step's."""
import csv
import glob
import os
import sys


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    names = [("check", "check_symmetric_pair_kernel"), ("resident", "pcg_resident_sym_kernel"), ("general", "pcg_cluster_kernel")]
    steps = []
    i = 0
    while i + 2 < len(rows):
        if all(names[j][1] in rows[i + j][2] for j in range(3)):
            steps.append(rows[i:i + 3])
            i += 3
        else:
            i += 1
    steps = [s for s in steps if s[1][1] - s[1][0] > 250_000][-50:]   # the 25-iteration steps
    med = lambda v: sorted(v)[len(v) // 2] / 1e3
    print(f"{len(steps)} steps")
    for j, (tag, _) in enumerate(names):
        print(f"  {tag:9s} {med([s[j][1] - s[j][0] for s in steps]):8.1f} us")
    print(f"  gap check -> resident   {med([s[1][0] - s[0][1] for s in steps]):8.1f} us")
    print(f"  gap resident -> general {med([s[2][0] - s[1][1] for s in steps]):8.1f} us")
    print(f"  first kernel start -> last kernel end {med([s[2][1] - s[0][0] for s in steps]):8.1f} us")


if __name__ == "__main__":
    main()
