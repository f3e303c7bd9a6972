"""Polynomial-commitment openings over one CRS verified together (PolyCommit::verify_batch over
bp_batch_axpy_s_vector): the MSM <h_poly.coeffs(), gs> of src/poly_commit/mod.rs:152-159 is paid once for the
batch, under verify_batch_aux's random linear combination (src/msm.rs:91-115).  Every opening of an accepted
batch is also accepted by the C oracle's restatement of poly_commit::verify; a corrupted or malformed member
rejects the batch."""
import numpy as np
import pytest

import bp_oracle as o
from test_gpu_configs import SEED, gen_points, is_identity, rand_scalars, xy8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [o.SECP256K1, o.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("k,n_coeffs,count,batch", [(10, 1024, 5, 5), (6, 50, 3, 8), (0, 1, 2, 2), (15, 1 << 15, 2, 3)])
def test_batch_of_openings(ctx, coracle, c, k, n_coeffs, count, batch):
    """2^15 generators: the provers' sessions replace their generators from the CRS's window tables; k = 0: a
    constant polynomial, no rounds; batch > count: openings repeated in a batch"""
    from bulletproofs_amd import host_mirror

    n = 1 << k
    rng = o.SplitMix64(SEED + 90 + k + c.id)
    threads = coracle.max_threads()
    G = gen_points(coracle, c, rng, n)
    hg = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c), xy8(c.g, c)])
    coeffs = rand_scalars(SEED + 91 + k, count * n_coeffs).reshape(count, n_coeffs, 4)
    r, x = rand_scalars(SEED + 92, count), rand_scalars(SEED + 93, count)
    res = host_mirror.poly_commit_verify_batch(ctx, c.id, G, hg, coeffs, r, x, batch=batch, seed_rand=15, seed_chal=16)
    assert res["ok_batch"] and res["ok_each"]
    for i in range(count):
        ores, _ = coracle.poly_verify(c.id, G, hg, res["commitment"][i], x[i], res["evaluation"][i], res["L"][i], res["R"][i],
                                      res["a"][i], res["r"][i], 16 + i, threads=threads)
        assert is_identity(ores), i
    if k <= 10:
        # the openings are the oracle prover's, message by message
        exp = coracle.poly_prove(c.id, G, hg, coeffs[count - 1], r[count - 1], x[count - 1], 15 + count - 1, 16 + count - 1, threads=threads)
        assert np.array_equal(res["L"][count - 1], exp["L"][:k]) and np.array_equal(res["a"][count - 1], exp["a"])
    for at in (0, batch - 1):
        bad = host_mirror.poly_commit_verify_batch(ctx, c.id, G, hg, coeffs, r, x, batch=batch, seed_rand=15, seed_chal=16, tamper=1,
                                                   tamper_at=at)
        assert not bad["ok_batch"] and not bad["ok_each"]
    if k:
        bad = host_mirror.poly_commit_verify_batch(ctx, c.id, G, hg, coeffs, r, x, batch=batch, seed_rand=15, seed_chal=16, tamper=2,
                                                   tamper_at=batch // 2)
        assert not bad["ok_batch"] and not bad["ok_each"]
