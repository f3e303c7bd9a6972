"""Arithmetic micro-benchmarks (field multiply, add/sub, XYZZ mixed add, Jacobian doubling) through
bp_microbench; prints rates per launch size.  usage (GPU box): python tools/microbench.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bulletproofs_amd as bp
from bulletproofs_amd._lib import lib
ctx = bp.Context(0)
print("peak %.3e" % ctx.int_mad_peak(2000))
def mb(i, blocks, iters):
    ms = C.c_float()
    rc = lib.bp_microbench(ctx.h, i, blocks, iters, C.byref(ms)); assert rc == 0
    return ms.value
for blocks in (256, 1024, 2048, 8192):
    it = 200
    ms = mb(0, blocks, it); muls = blocks*256*it*2
    print("fe_mul secp  blocks=%d: %.3f ms  %.3e mul/s  %.3e MAC/s" % (blocks, ms, muls/ms*1e3, muls*136/ms*1e3))
    ms = mb(4, blocks, it)
    print("fe_mul vesta blocks=%d: %.3f ms  %.3e mul/s  %.3e MAC/s" % (blocks, ms, muls/ms*1e3, muls*136/ms*1e3))
    ms = mb(1, blocks, it*10); ops = blocks*256*it*10*2
    print("add/sub      blocks=%d: %.3f ms  %.3e op/s" % (blocks, ms, ops/ms*1e3))
    ms = mb(2, blocks, 50); adds = blocks*128*50
    print("xyzz_madd    blocks=%d: %.3f ms  %.3e madd/s  %.3e MAC/s" % (blocks, ms, adds/ms*1e3, adds*1360/ms*1e3))
    ms = mb(3, blocks, 50)
    print("jac_dbl      blocks=%d: %.3f ms  %.3e dbl/s  %.3e MAC/s" % (blocks, ms, adds/ms*1e3, adds*7*136/ms*1e3))
