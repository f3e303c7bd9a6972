#!/usr/bin/env python3
"""bp_msm_resident over a registered base set with and without its wide window tables
(bp_bases_precompute_wide), 2^16 .. 2^20 points: wall time per call (host scalars: 32 bytes per point
cross PCIe in both forms), device phase and accumulation kernel, best of five.
usage (GPU box): python tools/time_wide.py [log2 of the largest size = 20] [log2 of the smallest = 16]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp  # noqa: E402
from bench import gen_points_on_gpu  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 20
low = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = bp.Context(0)
C = bp.SECP256K1
n_max = 1 << top
px, py = gen_points_on_gpu(ctx, n_max, 0)
rng = np.random.default_rng(3)
K = rng.integers(0, 1 << 64, (n_max, 4), dtype=np.uint64)
K[:, 3] &= np.uint64((1 << 63) - 1)
ch = K[:top].copy()
chi = ctx.batch_inversion(C, ch)
for logn in range(low, top + 1):
    n = 1 << logn
    bases = ctx.register_bases(C, (px[:n], py[:n], None))
    row = {"log2n": logn}
    for label in ("plain", "wide"):
        if label == "wide":
            t0 = time.perf_counter()
            bases.precompute_wide()
            row["table_build_ms"] = (time.perf_counter() - t0) * 1e3
        best = None
        for _ in range(6):
            t0 = time.perf_counter()
            out = bases.msm(K[:n])
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, ctx.last_kernel_ms(1), ctx.last_kernel_ms(0))
        row[label] = {"wall_ms": best[0] * 1e3, "device_ms": best[1], "accumulate_ms": best[2]}
        row[label + "_value"] = out[:2].tolist()
        # scalars built on the device: no PCIe in the call
        best = None
        for _ in range(6):
            t0 = time.perf_counter()
            bases.msm_s_vector(ch[:logn], chi[:logn])
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row[label]["s_vector_form_wall_ms"] = best * 1e3
    row["equal"] = row.pop("plain_value") == row.pop("wide_value")
    print(json.dumps(row), flush=True)
    bases.free()
