#!/usr/bin/env python3
"""Kernel launch sequence (start offset, duration, short name) from a rocprofv3 kernel trace CSV, for the
window around the first dispatch of <substring> that is longer than <min_us>.
    trace_sequence.py <kernel_trace.csv> <substring> <min_us> [count]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sub, min_us = sys.argv[2], float(sys.argv[3])
count = int(sys.argv[4]) if len(sys.argv) > 4 else 24
idx = next(i for i, r in enumerate(rows) if sub in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > min_us * 1e3)
t0 = int(rows[max(0, idx - count // 2)]["Start_Timestamp"])
for r in rows[max(0, idx - count // 2): idx + count // 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:90]}")
