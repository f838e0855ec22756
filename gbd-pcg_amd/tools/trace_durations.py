#!/usr/bin/env python3
"""Per-dispatch durations of selected kernels from a rocprofv3 kernel trace CSV, in launch order.
    trace_durations.py <kernel_trace.csv> <substring> [<substring> ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for name in sys.argv[2:]:
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if name in r["Kernel_Name"])
    print(name, len(d), "dispatches; durations in us, every 8th:", [round(x[1] / 1e3, 1) for x in d[::8]])
