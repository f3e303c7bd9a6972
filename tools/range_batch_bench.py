"""Proofs per second of batched aggregated range proofs (64 values x 64 bits, secp256k1) from one host
thread, for several batch sizes.  Usage: python tools/range_batch_bench.py [batch ...]"""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bulletproofs_amd import host_mirror  # noqa: E402
from bulletproofs_amd.api import Context  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]
    ctx = Context(0)
    m = n_bits = 64
    G, H, ugh, _, _ = host_mirror.range_bench_inputs(ctx, m, n_bits, 7)
    for B in batches:
        values = np.random.default_rng(B).integers(0, 1 << 64, (B, m), dtype=np.uint64)
        gamma = np.random.default_rng(B + 1).integers(0, 1 << 62, (B, m, 4), dtype=np.uint64)
        res = host_mirror.range_aggregate_prove_batch(ctx, 0, G, H, ugh, values, gamma, n_bits, reps=5, verify=(B == batches[0]))
        print(json.dumps({"batch": B, "batch_ms": round(res["batch_ms"], 3), "proofs_per_s": round(res["per_s"], 1), "ok": res["ok"]}),
              flush=True)


if __name__ == "__main__":
    main()
