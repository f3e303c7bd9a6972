#!/usr/bin/env python3
"""Prints the kernels of the last MSM of a rocprofv3 kernel trace as a timeline:
start/end in microseconds relative to the MSM's digits kernel, queue, name.
usage: python tools/trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "digits" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f  q%-3s %s" % (s / 1e3, e / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
