#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: polynomial commitment opening of degree 2^k - 1 (default
k = 20; benches/poly_commit_bench.rs) -- statement, prove, verify, lazy_verify and the fold
of two deferred checks -- through the GPU hot path with the seeded transcript stand-in.
Prints one JSON line.  usage: python bench_poly.py [log2_n]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402

SEED = 0xB0117E7 + 5
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
ctx = bp.Context(0)
p = (1 << 256) - (1 << 32) - 977
gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
r256 = 1 << 256
lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
gen = np.array(lim(gx * r256 % p) + lim(gy * r256 % p), dtype=np.uint64)
X = np.tile(gen[:4], (n + 1, 1))
Y = np.tile(gen[4:], (n + 1, 1))
rng = np.random.default_rng(SEED)


def rand_field(cnt):
    a = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 63) - 1)
    return a


px, py, _ = ctx.scale_points(bp.SECP256K1, (X, Y, None), rand_field(n + 1))  # CRS: random multiples of G
gs = (px[:n], py[:n], None)
hg = np.stack([np.concatenate([px[n], py[n]]), gen])
coeffs, r, x = rand_field(n), rand_field(1), rand_field(1)
host_mirror.poly_commit_roundtrip(ctx, bp.SECP256K1, gs, hg, coeffs, r, x, SEED, SEED + 1, halo=False)  # warm-up
res = host_mirror.poly_commit_roundtrip(ctx, bp.SECP256K1, gs, hg, coeffs, r, x, SEED, SEED + 1, halo=True)
assert res["verify_ok"] and res["lazy_ok"] and res["halo_ok"]
print(json.dumps({"metric": "poly_commit opening proofs/sec", "degree": n - 1, "curve": "secp256k1", "n_gpus": 1,
                  "prove_ms": res["prove_ms"], "verify_ms": res["verify_ms"], "prove_per_s": 1e3 / res["prove_ms"],
                  "verify_per_s": 1e3 / res["verify_ms"], "halo_fold_of_2_ok": res["halo_ok"],
                  "note": "seeded transcript stand-in; host buffers in, proof out; CRS resident on the GPU"}))
