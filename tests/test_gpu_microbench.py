"""The arithmetic self-checks behind bp_microbench: the plain-residue multiply of the MSM pipeline against its
portable form (id 8) and the batch-affine addition against the XYZZ formulas on multiples of the secp256k1
generator (id 12).  Both report the number of lanes that disagree."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bench_id,blocks,iters", [(8, 256, 40), (12, 64, 1)])
def test_arithmetic_self_checks(ctx, bench_id, blocks, iters):
    from bulletproofs_amd._lib import lib

    bad = C.c_float(-1.0)
    assert lib.bp_microbench(ctx.h, bench_id, blocks, iters, C.byref(bad)) == 0
    assert bad.value == 0.0


def test_timed_ids_run(ctx):
    from bulletproofs_amd._lib import lib

    for bench_id, blocks, iters in ((7, 64, 10), (11, 64, 32)):
        ms = C.c_float(-1.0)
        assert lib.bp_microbench(ctx.h, bench_id, blocks, iters, C.byref(ms)) == 0
        assert ms.value > 0.0
