#!/usr/bin/env python3
"""BASELINE config 4: aggregated range proof, m = 64 values x 64 bits (N = 4096), secp256k1
(benches/aggregate_range_bench.rs:93-97, i = 6), prove and verify through the GPU hot path
with the seeded transcript stand-in.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402

SEED = 0xB0117E7 + 4
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_bits = 64
n = m * n_bits
ctx = bp.Context(0)
p = (1 << 256) - (1 << 32) - 977
order = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
r = 1 << 256
lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
cnt = 2 * n + 3
X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (cnt, 1))
Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (cnt, 1))
rng = np.random.default_rng(SEED)
S = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
S[:, 3] &= np.uint64((1 << 63) - 1)
px, py, _ = ctx.scale_points(bp.SECP256K1, (X, Y, None), S)  # CRS: random multiples of G
gs = (px[:n], py[:n], None)
hs = (px[n:2 * n], py[n:2 * n], None)
ugh = np.concatenate([px[2 * n:], py[2 * n:]], axis=1)
values = rng.integers(0, 1 << 64, m, dtype=np.uint64)
gamma = rng.integers(0, 1 << 64, (m, 4), dtype=np.uint64)
gamma[:, 3] &= np.uint64((1 << 63) - 1)
host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, n_bits, SEED, SEED + 1, reps=1)
res = host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, n_bits, SEED, SEED + 1, reps=5)
assert res["ok"]
out = {"metric": "aggregated 64-bit range proofs/sec", "m": m, "n_bits": n_bits, "N": n,
       "prove_ms": res["prove_ms"], "verify_ms": res["verify_ms"],
       "prove_per_s": 1e3 / res["prove_ms"], "verify_per_s": 1e3 / res["verify_ms"], "n_gpus": 1,
       "note": "seeded transcript stand-in; host buffers in, proof out; *_ms and *_per_s are one proof at a time on "
               "one context, concurrent[] is W host threads with one context each on the same GPU"}
workers = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
out["concurrent"] = []
for w in workers:
    thr = host_mirror.range_aggregate_throughput(0, bp.SECP256K1, gs, hs, ugh, values, gamma, n_bits, w, 8, SEED, SEED + 1)
    assert thr["ok"]
    out["concurrent"].append({"workers": w, "prove_per_s": thr["prove_per_s"], "verify_per_s": thr["verify_per_s"]})
print(json.dumps(out))
