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
gs, hs, ugh, values, gamma = host_mirror.range_bench_inputs(ctx, m, n_bits, SEED)
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
# verify_batch at BATCH_SIZE 100 (benches/aggregate_range_bench.rs:21,65-84): four distinct proofs, cycled
rng = np.random.default_rng(SEED + 9)
bvals = rng.integers(0, 1 << 63, (4, m), dtype=np.uint64)
bgam = rng.integers(0, 1 << 64, (4, m, 4), dtype=np.uint64)
bgam[:, :, 3] &= np.uint64((1 << 63) - 1)
host_mirror.range_aggregate_verify_batch(ctx, bp.SECP256K1, gs, hs, ugh, bvals, bgam, n_bits, 8, SEED, SEED + 1)
vb = host_mirror.range_aggregate_verify_batch(ctx, bp.SECP256K1, gs, hs, ugh, bvals, bgam, n_bits, 100, SEED, SEED + 1)
assert vb["ok_device"] and vb["ok_host"] and vb["ok_direct"]
out["verify_batch_100"] = {"collect_ms": vb["collect_ms"], "device_merge_ms": vb["device_merge_ms"],
                           "host_merge_ms": vb["host_merge_ms"],
                           "batches_per_s_device_merge": 1e3 / (vb["collect_ms"] + vb["device_merge_ms"]),
                           "proofs_per_s_device_merge": 100e3 / (vb["collect_ms"] + vb["device_merge_ms"]),
                           "proofs_per_s_host_merge": 100e3 / (vb["collect_ms"] + vb["host_merge_ms"]),
                           "direct_ms": vb["direct_ms"], "proofs_per_s_direct": 100e3 / vb["direct_ms"],
                           "note": "collect = 100 x verify_aux (scalars of the deferred MSM, one at a time on one context); "
                                   "merge = the random linear combination and its single MSM; direct = the whole batch with every proof's "
                                   "CRS scalars built and added on the device (bp_batch_axpy_range_agg), one MSM"}
print(json.dumps(out))
