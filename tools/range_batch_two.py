"""Two (or more) batches of range proofs in flight: `threads` host threads, each with a context of its own on GPU 0,
prove batches of `batch` proofs at the same time.  Usage: python tools/range_batch_two.py [batch] [threads] [reps]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bulletproofs_amd import host_mirror  # noqa: E402
from bulletproofs_amd.api import Context  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    m = n_bits = 64
    ctxs = [Context(0) for _ in range(T)]
    G, H, ugh, _, _ = host_mirror.range_bench_inputs(ctxs[0], m, n_bits, 7)
    values = np.random.default_rng(B).integers(0, 1 << 64, (B, m), dtype=np.uint64)
    gamma = np.random.default_rng(B + 1).integers(0, 1 << 62, (B, m, 4), dtype=np.uint64)
    for c in ctxs:  # warm-up: workspaces
        host_mirror.range_aggregate_prove_batch(c, 0, G, H, ugh, values, gamma, n_bits, reps=1)
    out = [None] * T

    def body(k):
        out[k] = host_mirror.range_aggregate_prove_batch(ctxs[k], 0, G, H, ugh, values, gamma, n_bits, reps=reps)

    th = [threading.Thread(target=body, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    # each call sets up its CRS and runs one untimed batch first: count the timed batches only, over their own clocks
    per = [o["batch_ms"] for o in out]
    print(json.dumps({"batch": B, "threads": T, "batch_ms_each": [round(x, 2) for x in per],
                      "proofs_per_s_sum": round(sum(B / x * 1e3 for x in per), 1), "wall_s_with_setup": round(dt, 3)}))


if __name__ == "__main__":
    main()
