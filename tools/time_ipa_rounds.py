#!/usr/bin/env python3
"""per-call wall time of the IPA session at small n."""
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
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 12
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
X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (2 * n + 1, 1))
Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (2 * n + 1, 1))
px, py, _ = ctx.scale_points(C, (X, Y, None), rand_field(2 * n + 1))
bases = ctx.register_bases(C, (px[:2 * n + 1], py[:2 * n + 1], None))
use_tables = os.environ.get("BP_TABLES") == "1"
if use_tables:
    bases.precompute()
u = np.concatenate([px[2 * n], py[2 * n]])
ch = rand_field(logn)
chi = ctx.batch_inversion(C, ch)
cs = [int(v) for v in sys.argv[2:]] or [0, 0, 0]
for c_win in cs:
    ctx.set_msm_window(c_win)
    print("window", c_win, end="  ")
    s = ctx.ipa_begin_resident(bases, 0, 0, n, rand_field(n), rand_field(n))
    if use_tables:
        s.set_u_resident(2 * n, rand_field(1))
    else:
        s.set_u(u)
    t_lr, t_fold, dev = [], [], []
    for j in range(logn):
        t0 = time.perf_counter()
        s.round_lr()
        t_lr.append(time.perf_counter() - t0)
        dev.append(ctx.last_kernel_ms(1))
        t0 = time.perf_counter()
        s.round_fold(ch[j:j + 1], chi[j:j + 1])
        t_fold.append(time.perf_counter() - t0)
    # the fully folded generator equals MSM(gs, s-vector) (src/poly_commit/mod.rs:108-113)
    G, H, Af, Bf = s.read()
    ss = ctx.s_vector(C, ch, chi)
    exp = ctx.msm_unchecked(C, (px[:n], py[:n], None), ss)
    ex, ey, _ = ctx.normalize_batch(C, exp.reshape(1, 12))
    assert np.array_equal(G[0], ex) and np.array_equal(G[1], ey), "folded generator mismatch"
    s.free()
    print("folds (ms):", " ".join("%.3f" % (1e3 * t) for t in t_fold))
    print("round_lr (ms):", " ".join("%.3f" % (1e3 * t) for t in t_lr))
    print("n=2^%d: round_lr wall %.3f ms (device phase %.3f ms), round_fold %.3f ms" % (
        logn, 1e3 * np.median(t_lr), np.median(dev), 1e3 * np.median(t_fold)), flush=True)
