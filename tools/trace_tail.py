#!/usr/bin/env python3
"""Prints kernels [first, last) (indices counted from the end when negative) of a rocprofv3 kernel
trace as a timeline: start in microseconds relative to the first one printed, duration, the idle gap
before the kernel, and its name.
usage: python tools/trace_tail.py <kernel_trace.csv> [first] [last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else -150
last = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
sel = rows[first:last]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
