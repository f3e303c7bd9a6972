#!/usr/bin/env python3
"""The streaming kernels alone at 2^k elements (fold_scalars, inner_product, s_vector), five launches
each, inputs resident in HBM: the program tools/prof_stream.sh runs under rocprofv3 for their memory-side
byte counts.  Prints one JSON line per kernel with the HIP-event duration.
usage (GPU box): python tools/stream_kernels.py [k=24]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd._lib import lib  # noqa: E402

ctx = bp.Context(0)
C = bp.SECP256K1
rng = np.random.default_rng(7)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def rand_field(k):
    a = rng.integers(0, 1 << 64, (k, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 63) - 1)
    return a


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        fn()
        ts.append(ctx.last_kernel_ms(2))
    return min(ts)


a, b, out = dev(rand_field(n)), dev(rand_field(n)), dev(np.zeros((n, 4), dtype=np.uint64))
x, y = rand_field(1), rand_field(1)
ms = best(lambda: ctx._chk(lib.bp_fold_scalars_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, x.ctypes.data, y.ctypes.data, out.data_ptr())))
print(json.dumps({"kernel": "fold_scalars", "n": n, "ms": ms, "algorithmic_bytes": 96 * n, "GBps": 96 * n / ms / 1e6}), flush=True)
res = np.zeros(4, dtype=np.uint64)
ms = best(lambda: ctx._chk(lib.bp_inner_product_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, res.ctypes.data)))
print(json.dumps({"kernel": "inner_product", "n": n, "ms": ms, "algorithmic_bytes": 64 * n, "GBps": 64 * n / ms / 1e6}), flush=True)
ch = rand_field(logn)
chi = ctx.batch_inversion(C, ch)
ms = best(lambda: ctx._chk(lib.bp_s_vector_dev(ctx.h, C, ch.ctypes.data, chi.ctypes.data, logn, out.data_ptr())))
print(json.dumps({"kernel": "s_vector", "n": n, "ms": ms, "algorithmic_bytes": 32 * n, "GBps": 32 * n / ms / 1e6}), flush=True)
