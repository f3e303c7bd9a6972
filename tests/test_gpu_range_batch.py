"""Batches of aggregated range proofs (bp_range_batch_*, RangeAggregate::prove_batch): B proofs of one
shape share every launch of src/range/aggregate.rs:23-170 and of the argument's rounds
(src/ipa/mod.rs:38-97).  Every proof of a batch must equal, message by message, the proof the C oracle's
restatement of the reference's prover returns for that proof's values, blinders and challenges."""
import numpy as np
import pytest

import bp_oracle as o
from test_gpu_configs import SEED, assert_range_proof_equal, gen_points, is_identity, rand_scalars, xy8

pytestmark = pytest.mark.gpu


def setup(coracle, c, m, n_bits, batch, seed):
    n = m * n_bits
    rng = o.SplitMix64(seed)
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    ugh = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c) for _ in range(3)])
    values = np.random.default_rng(seed).integers(0, 1 << n_bits, (batch, m), dtype=np.uint64)
    values[0, 0] = (1 << n_bits) - 1
    values[-1, -1] = 0
    gamma = rand_scalars(seed + 1, batch * m).reshape(batch, m, 4)
    return G, H, ugh, values, gamma


@pytest.mark.parametrize("c", [o.SECP256K1, o.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("m,n_bits,batch", [(64, 64, 16), (8, 32, 5), (1, 64, 64), (4, 16, 128), (2, 1, 3), (1, 8, 1)])
def test_every_proof_of_a_batch_equals_the_oracles(ctx, coracle, c, m, n_bits, batch):
    """(64, 64, 16) is the shape of benches/aggregate_range_bench.rs; the others: a batch that is not a
    power of two, the single-value proof at a large batch, the largest batch, N = 2 (one round), a batch of one"""
    from bulletproofs_amd import host_mirror

    if c is o.VESTA and m == 64:
        pytest.skip("the 64 x 64 case is run on the curve the reference benches use")
    G, H, ugh, values, gamma = setup(coracle, c, m, n_bits, batch, SEED + 70 + m + batch)
    threads = coracle.max_threads()
    res = host_mirror.range_aggregate_prove_batch(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=51, seed_chal=52, verify=True)
    assert res["ok"]
    for b in range(batch):
        exp = coracle.range_agg_prove(c.id, G, H, ugh, values[b], gamma[b], n_bits, 51 + b, 52 + b, threads=threads)
        assert_range_proof_equal(res["proofs"][b], exp)
    # the oracle's verifier accepts the first and the last proof
    for b in (0, batch - 1):
        pr = res["proofs"][b]
        ov = coracle.range_agg_verify(c.id, G, H, ugh, values[b], gamma[b], n_bits, pr["points"], pr["scalars"], pr["lr"],
                                      (51 + b) ^ 0x5EED, 52 + b, threads=threads)
        assert is_identity(ov)


def test_uploaded_blinders_give_the_same_proofs(ctx, coracle):
    """the blinders drawn by the host generator and uploaded (what a caller with its own generator does)
    against the device's draw of the same streams"""
    from bulletproofs_amd import host_mirror

    c = o.SECP256K1
    G, H, ugh, values, gamma = setup(coracle, c, 4, 16, 6, SEED + 71)
    dev = host_mirror.range_aggregate_prove_batch(ctx, c.id, G, H, ugh, values, gamma, 16, seed_rand=61, seed_chal=62)
    up = host_mirror.range_aggregate_prove_batch(ctx, c.id, G, H, ugh, values, gamma, 16, seed_rand=61, seed_chal=62, host_blinders=True)
    for a, b in zip(dev["proofs"], up["proofs"]):
        assert_range_proof_equal(a, b)
    exp = coracle.range_agg_prove(c.id, G, H, ugh, values[5], gamma[5], 16, 61 + 5, 62 + 5, threads=2)
    assert_range_proof_equal(up["proofs"][5], exp)


def test_batch_object_rejects_bad_shapes_and_order(ctx, coracle):
    from bulletproofs_amd import _lib
    from bulletproofs_amd.api import Bases
    import ctypes as C

    c = o.SECP256K1
    rng = o.SplitMix64(SEED + 72)
    n = 16
    x, y, _ = gen_points(coracle, c, rng, 2 * n + 3)
    bases = Bases(ctx, c.id, (x, y, None))
    rb = C.c_void_p()
    # no window tables yet
    assert _lib.lib.bp_range_batch_begin(ctx.h, bases.h, 4, 2, 8, C.byref(rb)) == _lib.BP_E_STATE
    bases.precompute()
    assert _lib.lib.bp_range_batch_begin(ctx.h, bases.h, 129, 2, 8, C.byref(rb)) == _lib.BP_E_ARG     # batch too large
    assert _lib.lib.bp_range_batch_begin(ctx.h, bases.h, 4, 3, 8, C.byref(rb)) == _lib.BP_E_ARG       # N not a power of two
    assert _lib.lib.bp_range_batch_begin(ctx.h, bases.h, 4, 4, 8, C.byref(rb)) == _lib.BP_E_ARG       # base set too short
    assert _lib.lib.bp_range_batch_begin(ctx.h, bases.h, 4, 2, 8, C.byref(rb)) == _lib.BP_OK
    try:
        assert _lib.lib.bp_range_batch_size(rb) == 4
        out = np.zeros((4, 2, 12), dtype=np.uint64)
        sc = np.zeros((4, 4), dtype=np.uint64)
        # phases out of order
        assert _lib.lib.bp_range_batch_eval(rb, sc.ctypes.data, out.ctypes.data) == _lib.BP_E_STATE
        assert _lib.lib.bp_range_batch_ipa_begin(rb, sc.ctypes.data) == _lib.BP_E_STATE
        assert _lib.lib.bp_range_batch_round_lr(rb, out.ctypes.data) == _lib.BP_E_STATE
        assert _lib.lib.bp_range_batch_finish(rb, out.ctypes.data) == _lib.BP_E_STATE
        assert _lib.lib.bp_range_batch_ipa_len(rb) == 0
    finally:
        _lib.lib.bp_range_batch_free(rb)


@pytest.mark.parametrize("rounds", ["0", "1", "2", "3"])
def test_generator_replacement_inside_a_batch(ctx, coracle, rounds, monkeypatch):
    """the sessions of a batch replace their generators by the folded ones after `rounds` rounds (0: never): the
    proofs do not depend on when, or whether, that happens"""
    from bulletproofs_amd import host_mirror

    monkeypatch.setenv("BP_RANGE_BATCH_REBASE", rounds)
    for c, m, n_bits, batch in ((o.SECP256K1, 16, 32, 6), (o.VESTA, 8, 32, 4)):
        G, H, ugh, values, gamma = setup(coracle, c, m, n_bits, batch, SEED + 73 + int(rounds))
        res = host_mirror.range_aggregate_prove_batch(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=71, seed_chal=72, verify=True)
        assert res["ok"]
        for b in (0, batch - 1):
            exp = coracle.range_agg_prove(c.id, G, H, ugh, values[b], gamma[b], n_bits, 71 + b, 72 + b, threads=coracle.max_threads())
            assert_range_proof_equal(res["proofs"][b], exp)
