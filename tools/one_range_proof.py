#!/usr/bin/env python3
"""One aggregated range proof at a time (m values x 64 bits), for kernel traces: prints prove/verify ms."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402

SEED = 0xB0117E7 + 4
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = bp.Context(0)
gs, hs, ugh, values, gamma = host_mirror.range_bench_inputs(ctx, m, 64, SEED)
host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, 64, SEED, SEED + 1, reps=1)
res = host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, 64, SEED, SEED + 1, reps=reps)
assert res["ok"]
print("m=%d prove_ms %.3f verify_ms %.3f" % (m, res["prove_ms"], res["verify_ms"]))
