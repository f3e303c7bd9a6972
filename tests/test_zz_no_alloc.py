"""GPU test of bp_inner_product_dev_noalloc, the entry point for callers that cannot allocate:
through the C ABI its device output equals bp_inner_product_dev's (bit-exact), on two calls in a
row, the second with a smaller input (the static scratch array keeps the first call's partial
sums), plus the empty input and the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_no_alloc_inner_product_matches_the_allocating_entry_point(ctx):
    import torch

    from bulletproofs_amd import _lib

    lib = _lib.lib
    rng = np.random.default_rng(20260211)
    for curve in (0, 1):
        # any value below 2^254 is the Montgomery form of some element of either scalar field
        n1, n2 = (1 << 20) + 77, 333
        a = rng.integers(0, 1 << 64, (n1, 4), dtype=np.uint64)
        b = rng.integers(0, 1 << 64, (n1, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 62) - 1)
        b[:, 3] &= np.uint64((1 << 62) - 1)
        da = torch.from_numpy(a.view(np.int64)).cuda()
        db = torch.from_numpy(b.view(np.int64)).cuda()
        d_out = torch.full((4,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for n in (n1, n2, 0):
            want = np.zeros(4, dtype=np.uint64)
            assert lib.bp_inner_product_dev(ctx.h, curve, da.data_ptr(), db.data_ptr(), n, want.ctypes.data) == 0
            assert lib.bp_inner_product_dev_noalloc(ctx.h, curve, da.data_ptr(), db.data_ptr(), n, d_out.data_ptr()) == 0
            assert lib.bp_sync(ctx.h) == 0
            got = d_out.cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), (curve, n)
            if n == 0:
                assert not got.any()
    # sizes beyond the stated maximum are refused, with a message
    assert lib.bp_inner_product_dev_noalloc(ctx.h, 0, da.data_ptr(), db.data_ptr(), (1 << 28) + 1, d_out.data_ptr()) == _lib.BP_E_ARG
    assert b"BP_NOALLOC_MAX_N" in lib.bp_last_error(ctx.h)
    assert lib.bp_inner_product_dev_noalloc(ctx.h, 0, da.data_ptr(), db.data_ptr(), 8, None) == _lib.BP_E_ARG
