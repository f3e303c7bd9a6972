"""A polynomial-commitment opening of degree 2^log2n - 1 over a resident CRS (benches/poly_commit_bench.rs shape),
twice; prints the second run's prove and verify times.  A target for rocprofv3 and BP_HOST_TRACE=1.
Usage: python tools/poly_prove.py [log2n]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << lg
    C = bp.SECP256K1
    ctx = bp.Context(0)
    px, py = bench.gen_points_on_gpu(ctx, n, 0)
    rng = np.random.default_rng(6)

    def rand_field(cnt):
        a = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 63) - 1)
        return a

    hx, hy, _ = ctx.scale_points(C, (px[:2], py[:2], None), rand_field(2))
    hg = np.stack([np.concatenate([hx[0], hy[0]]), np.concatenate([hx[1], hy[1]])])
    gs = (px, py, None)
    coeffs, r, x = rand_field(n), rand_field(1), rand_field(1)
    host_mirror.poly_commit_roundtrip(ctx, C, gs, hg, coeffs, r, x, 1, 2, halo=False)
    print("---- second run ----", file=sys.stderr, flush=True)
    res = host_mirror.poly_commit_roundtrip(ctx, C, gs, hg, coeffs, r, x, 1, 2, halo=False)
    print(json.dumps({"log2n": lg, "prove_ms": res["prove_ms"], "verify_ms": res["verify_ms"], "ok": bool(res["verify_ok"] and res["lazy_ok"])}))


if __name__ == "__main__":
    main()
