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


def test_no_alloc_calls_on_two_contexts_of_one_gpu_do_not_share_partial_sums():
    """Every context owns a slice of the static array: calls enqueued back to back on two contexts
    (two non-blocking streams of one GPU, nothing orders them) each return their own inner product."""
    import torch

    import bulletproofs_amd as bp
    from bulletproofs_amd import _lib

    lib = _lib.lib
    rng = np.random.default_rng(20260212)
    n = (1 << 21) + 5
    vecs = []
    for _ in range(4):
        v = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 62) - 1)
        vecs.append(torch.from_numpy(v.view(np.int64)).cuda())
    c1, c2 = bp.Context(0), bp.Context(0)
    try:
        want = [np.zeros(4, dtype=np.uint64) for _ in range(2)]
        assert lib.bp_inner_product_dev(c1.h, 0, vecs[0].data_ptr(), vecs[1].data_ptr(), n, want[0].ctypes.data) == 0
        assert lib.bp_inner_product_dev(c2.h, 0, vecs[2].data_ptr(), vecs[3].data_ptr(), n, want[1].ctypes.data) == 0
        out = torch.full((2, 4), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for rep in range(8):
            out.fill_(-1)
            torch.cuda.synchronize()
            # neither call waits: the launches of the two contexts are in flight together
            assert lib.bp_inner_product_dev_noalloc(c1.h, 0, vecs[0].data_ptr(), vecs[1].data_ptr(), n, out[0].data_ptr()) == 0
            assert lib.bp_inner_product_dev_noalloc(c2.h, 0, vecs[2].data_ptr(), vecs[3].data_ptr(), n, out[1].data_ptr()) == 0
            assert lib.bp_sync(c1.h) == 0 and lib.bp_sync(c2.h) == 0
            got = out.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), rep
    finally:
        c1.close()
        c2.close()
