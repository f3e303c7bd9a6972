#!/usr/bin/env python3
"""Calibration run for the memory-side counters in the MSM gather pattern: one scattered
read of every 64-byte record of a 2^24-record (1 GiB) table.  Run under
`rocprofv3 --pmc FETCH_SIZE`; the known byte count of k_mb_gather64 is 64 * 2^24."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd._lib import lib  # noqa: E402

ctx = bp.Context(0)
ms = C.c_float()
log2_records = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx._chk(lib.bp_microbench(ctx.h, 5, 1, log2_records, C.byref(ms)))
nbytes = 64 << log2_records
print("gather64: %d records, %.3f ms, %.1f GB/s of record bytes" % (1 << log2_records, ms.value, nbytes / ms.value / 1e6))
