#!/usr/bin/env python3
"""per-round wall time of a polynomial-commitment session (src/poly_commit/mod.rs:69-105) over 2^k
generators registered on the GPU: round_lr and round_fold of every round.
usage (GPU box): python tools/time_poly_rounds.py [k=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bp  # noqa: E402

ctx = bp.Context(0)
C = bp.SECP256K1
rng = np.random.default_rng(7)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn


def rand_field(k):
    a = rng.integers(0, 1 << 64, (k, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 63) - 1)
    return a


p = (1 << 256) - (1 << 32) - 977
gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
r = 1 << 256
lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (n + 1, 1))
Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (n + 1, 1))
px, py, _ = ctx.scale_points(C, (X, Y, None), rand_field(n + 1))
bases = ctx.register_bases(C, (px[:n], py[:n], None))
u = np.concatenate([px[n], py[n]])
ch = rand_field(logn)
chi = ctx.batch_inversion(C, ch)
for rep in range(2):
    va, vb = rand_field(n), rand_field(n)
    t00 = time.perf_counter()
    s = ctx.ipa_begin_resident(bases, bp.IPA_POLYCOMMIT, 0, 0, va, vb)
    s.set_u(u)
    t_begin = time.perf_counter() - t00
    t_lr, t_fold = [], []
    for j in range(logn):
        t0 = time.perf_counter()
        s.round_lr()
        t_lr.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        s.round_fold(ch[j:j + 1], chi[j:j + 1])
        t_fold.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    s.read()
    t_read = time.perf_counter() - t0
    s.free()
    print("begin %.2f ms  read %.2f ms  total rounds %.2f ms" % (t_begin * 1e3, t_read * 1e3, (sum(t_lr) + sum(t_fold)) * 1e3))
    print("lr  ", " ".join("%.2f" % (t * 1e3) for t in t_lr))
    print("fold", " ".join("%.2f" % (t * 1e3) for t in t_fold))
