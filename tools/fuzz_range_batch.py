"""Randomised parity run of batched range proving (bp_range_batch_*, RangeAggregate::prove_batch) against the C
oracle: random curve, m values x n_bits bits, batch size; some proofs of every batch (the first, the last and a
few between) are compared with the oracle's proofs message by message, and the library's own verifier checks
all of them.  Stops at the first mismatch (exit 1) or after the time given.
Usage (GPU box): python tools/fuzz_range_batch.py [seconds] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bp_oracle as o  # noqa: E402
import c_oracle  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402
from test_gpu_configs import assert_range_proof_equal  # noqa: E402
from test_gpu_range_batch import setup  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    coracle = c_oracle.COracle()
    threads = coracle.max_threads()
    ctx = bp.Context(0)
    t_end = time.time() + budget
    cases = 0
    while time.time() < t_end:
        c = (o.SECP256K1, o.VESTA)[int(rng.integers(0, 2))]
        m = 1 << int(rng.integers(0, 7))
        n_bits = 1 << int(rng.integers(0, 7))
        n = m * n_bits
        if n < 2:
            continue
        cap = max(1, min(128, 200000 // n))
        batch = int(rng.integers(1, cap + 1)) if rng.random() < 0.7 else cap
        s0 = int(rng.integers(1, 1 << 30))
        G, H, ugh, values, gamma = setup(coracle, c, m, n_bits, batch, s0)
        sr, sc = int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 20))
        res = host_mirror.range_aggregate_prove_batch(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=sr, seed_chal=sc, verify=True)
        case = {"curve": c.name, "m": m, "n_bits": n_bits, "batch": batch, "setup_seed": s0, "seed_rand": sr, "seed_chal": sc}
        if not res["ok"]:
            print(json.dumps({"verify_failed": True, **case}), flush=True)
            sys.exit(1)
        picks = sorted(set([0, batch - 1] + [int(v) for v in rng.integers(0, batch, 3)]))
        for b in picks:
            exp = coracle.range_agg_prove(c.id, G, H, ugh, values[b], gamma[b], n_bits, sr + b, sc + b, threads=threads)
            try:
                assert_range_proof_equal(res["proofs"][b], exp)
            except AssertionError:
                print(json.dumps({"mismatch": True, "proof": b, **case}), flush=True)
                sys.exit(1)
        cases += 1
        if cases % 10 == 0:
            print(json.dumps({"cases": cases, "last": case}), flush=True)
    print(json.dumps({"ok": True, "cases": cases, "seed": seed}))


if __name__ == "__main__":
    main()
