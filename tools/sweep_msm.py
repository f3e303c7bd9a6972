#!/usr/bin/env python3
"""Tuning sweep: MSM device time over window width and
reduction segment length.  usage: [SWEEP_C=8,16] python tools/sweep_msm.py [log2n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bulletproofs_amd as bp  # noqa: E402

sys.argv = sys.argv[:1] + [a for a in sys.argv[1:]]
logs = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20]
ctx = bp.Context(0)
C = bp.VESTA if os.environ.get("CURVE") == "vesta" else bp.SECP256K1
rng = np.random.default_rng(7)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def rand_field(n):
    a = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << (61 if C == bp.VESTA else 63)) - 1)
    return a


def gen_points(n):
    p = (1 << 256) - (1 << 32) - 977
    order = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    if C == bp.VESTA:
        p = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
        order = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
        gx, gy = p - 1, 2
    r = 1 << 256
    lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
    X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (n, 1))
    Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (n, 1))
    rm = r % order
    vals = [((i + 1) * rm) % order for i in range(n)]
    S = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        S[:, k] = np.array([(v >> (64 * k)) & ((1 << 64) - 1) for v in vals], dtype=np.uint64)
    ox, oy, _ = ctx.scale_points(C, (X, Y, None), S)
    return ox, oy


big = 1 << max(logs)
bx, by = gen_points(1 << 14)
reps = max(1, big // (1 << 14))
BX, BY = dev(np.tile(bx, (reps, 1))), dev(np.tile(by, (reps, 1)))
K = dev(rand_field(big))
for logn in logs:
    n = 1 << logn
    for c in ([int(v) for v in os.environ['SWEEP_C'].split(',')] if os.environ.get('SWEEP_C') else range(max(4, logn - 7), 17)):
        row = []
        for seg in ((0,) if os.environ.get('SWEEP_C') else (0, 4, 8, 16)):
            if seg:
                os.environ["BP_MSM_SEG_LEN"] = str(seg)
            else:
                os.environ.pop("BP_MSM_SEG_LEN", None)
            ctx.set_msm_window(c)
            ts = []
            for _ in range(3):
                ctx.msm_dev(C, BX.data_ptr(), BY.data_ptr(), None, K.data_ptr(), n)
                ts.append(ctx.last_kernel_ms(1))
            row.append(min(ts))
        print(f"log2n={logn} c={c:2d}  " + "  ".join(f"{v:.3f}" for v in row), flush=True)
