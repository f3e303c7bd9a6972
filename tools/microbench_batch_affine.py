"""Batch-affine additions (bp_microbench id 11, self-check id 12) against the XYZZ mixed addition (id 7), all on
the secp256k1 base field in the pipeline's plain form.  usage (GPU box): python tools/microbench_batch_affine.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd._lib import lib  # noqa: E402

ctx = bp.Context(0)


def mb(i, blocks, iters):
    ms = C.c_float()
    rc = lib.bp_microbench(ctx.h, i, blocks, iters, C.byref(ms))
    assert rc == 0, rc
    return ms.value


print("self-check: lanes that disagree =", mb(12, 64, 1))
ctx.int_mad_peak(2000)
for blocks in (1024, 2048, 4096):
    adds = blocks * 128 * 200
    ms = min(mb(7, blocks, 200) for _ in range(3))
    print("xyzz_madd (registers)        blocks=%d: %.3f ms  %.3e additions/s" % (blocks, ms, adds / ms * 1e3))
    for t in (64, 256, 1024):
        if blocks * 128 * t * 32 > (8 << 30):
            continue
        adds = blocks * 128 * t
        ms = min(mb(11, blocks, t) for _ in range(3))
        print("batch-affine T=%-4d          blocks=%d: %.3f ms  %.3e additions/s  (%.0f MB parked)" %
              (t, blocks, ms, adds / ms * 1e3, adds * 32 / 1e6))
