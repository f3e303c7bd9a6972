"""GPU tests of the array-of-structures entry points (72-byte affine records): same results as the
plane forms, and as the C oracle."""
import numpy as np
import pytest

import bp_oracle as o
from test_gpu_parity import random_scalars, seeded_points

pytestmark = pytest.mark.gpu

CURVES = [o.SECP256K1, o.VESTA]
ids = lambda c: c.name  # noqa: E731


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [0, 1, 7, 600, 5000, 70000])
def test_msm_aos_equals_planes(ctx, coracle, c, n):
    from bulletproofs_amd.api import pack_aos

    px, py = seeded_points(coracle, c, max(n, 1), 13, 17)
    px, py = px[:n], py[:n]
    inf = np.zeros(n, dtype=np.uint8)
    inf[::5] = 1  # identities among the bases: x, y of those records are ignored
    _, K = random_scalars(c, max(n, 1), n + 1)
    K = K[:n]
    rec = pack_aos((px, py, inf))
    # padding bytes and the coordinates of identity records carry junk in a real Vec<Affine>
    rec[:, 65:] = 0xA5
    rec[inf != 0, :64] = 0x5A
    got = ctx.msm_aos(c.id, rec, K)
    assert np.array_equal(got, ctx.msm_unchecked(c.id, (px, py, inf), K))
    if n:
        assert np.array_equal(got, coracle.msm(c.id, (px, py, inf), K, 8))


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_bases_register_aos(ctx, coracle, c):
    from bulletproofs_amd.api import pack_aos

    n = 3000
    px, py = seeded_points(coracle, c, n, 3, 29)
    inf = np.zeros(n, dtype=np.uint8)
    inf[7] = inf[2999] = 1
    _, K = random_scalars(c, n, 8)
    bases = ctx.register_bases_aos(c.id, pack_aos((px, py, inf)))
    assert len(bases) == n
    assert np.array_equal(bases.msm(K), coracle.msm(c.id, (px, py, inf), K, 8))
    # a slice of the set, as crs.gs[0..n] (src/ipa/types.rs:81-82)
    assert np.array_equal(bases.msm(K[:1000], offset=500), coracle.msm(c.id, (px[500:1500], py[500:1500], inf[500:1500]), K[:1000], 8))
    bases.free()
