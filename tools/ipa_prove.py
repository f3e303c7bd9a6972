"""The 16 rounds of an inner-product argument over 2 x 2^16 resident generators (benches/ipa_bench.rs shape),
a few times; prints per-round wall times of the last run.  A target for rocprofv3.
Usage: python tools/ipa_prove.py [log2n] [runs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = 1 << lg
    C = bp.SECP256K1
    ctx = bp.Context(0)
    px, py = bench.gen_points_on_gpu(ctx, 2 * n + 1, 0)
    rng = np.random.default_rng(5)

    def rand_field(cnt):
        a = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 63) - 1)
        return a

    ch = rand_field(lg)
    chi = ctx.batch_inversion(C, ch)
    bases = ctx.register_bases(C, (px, py, None))
    if os.environ.get("BP_TOOL_NO_TABLES") is None:
        t0 = time.perf_counter()
        bases.precompute()
        ctx.sync()
        print(json.dumps({"window_tables_ms": (time.perf_counter() - t0) * 1e3}), flush=True)
    u_xy = np.concatenate([px[2 * n], py[2 * n]])
    wa, wb = rand_field(n), rand_field(n)
    for _ in range(runs):
        s = ctx.ipa_begin_resident(bases, bp.IPA_BULLETPROOF, 0, n, wa, wb)
        s.set_u(u_xy)
        rounds = []
        t0 = time.perf_counter()
        for j in range(lg):
            ta = time.perf_counter()
            s.round_lr()
            tb = time.perf_counter()
            s.round_fold(ch[j:j + 1], chi[j:j + 1])
            tc = time.perf_counter()
            rounds.append((round((tb - ta) * 1e3, 3), round((tc - tb) * 1e3, 3)))
        ctx.sync()
        dt = time.perf_counter() - t0
        s.free()
    print(json.dumps({"log2n": lg, "prove_rounds_ms": dt * 1e3, "rounds_lr_fold_ms": rounds}))


if __name__ == "__main__":
    main()
