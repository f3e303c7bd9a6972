"""Montgomery vs plain-residue secp256k1 base-field multiply and XYZZ mixed addition, through
bp_microbench (ids 0/6, 2/7), after the plain multiply's self-check (id 8).
usage (GPU box): python tools/microbench_plain.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bulletproofs_amd as bp
from bulletproofs_amd._lib import lib
ctx = bp.Context(0)
def mb(i, blocks, iters):
    ms = C.c_float()
    rc = lib.bp_microbench(ctx.h, i, blocks, iters, C.byref(ms)); assert rc == 0, rc
    return ms.value
print("self-check: lanes that disagree =", mb(8, 2048, 50))
print("peak %.3e MAC/s" % ctx.int_mad_peak(2000))
for blocks in (1024, 2048, 8192):
    it = 200
    muls = blocks * 256 * it * 2
    for name, i in (("montgomery", 0), ("plain", 6)):
        ms = min(mb(i, blocks, it) for _ in range(3))
        print("fe_mul %-10s blocks=%d: %.3f ms  %.3e mul/s" % (name, blocks, ms, muls / ms * 1e3))
    adds = blocks * 128 * 50
    for name, i in (("montgomery", 2), ("plain", 7)):
        ms = min(mb(i, blocks, 50) for _ in range(3))
        print("xyzz_madd %-10s blocks=%d: %.3f ms  %.3e madd/s" % (name, blocks, ms, adds / ms * 1e3))
